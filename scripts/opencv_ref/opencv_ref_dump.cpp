// opencv_ref_dump: runs the OpenCV 4.8 calls the reference makes on the stabilization hot path on the inputs `opencv_ref_compare.py
// export` wrote, and dumps their outputs for `opencv_ref_compare.py compare` to hold the oracle (and through it the HIP kernels) against.
//
// NOT part of the product, the oracle or the test suite: this image has no OpenCV, so the file is never compiled
// here and nothing depends on it.  It exists so that anyone WITH OpenCV 4.8.0 (the reference's pin, Scripts/setup_deb.sh:42) can close the
// rows whose arithmetic lives in that library -- a4 / a5 / a7 / a9 / a12 / a14 -- against the real thing:
//
//   g++ -O2 -std=c++17 opencv_ref_dump.cpp -o opencv_ref_dump $(pkg-config --cflags --libs opencv4)
//   (with Eigen 3.4, the reference's pin -- Scripts/setup_deb.sh:133 --, add  -DLVK_WITH_EIGEN $(pkg-config --cflags eigen3)  : row a10 as well;
//    scripts/opencv_ref/run.sh does all of it in one command)
//   python scripts/opencv_ref/opencv_ref_compare.py export /tmp/lvk_cv          # inputs, from tests/golden + the clip generator
//   ./opencv_ref_dump /tmp/lvk_cv                                                # <name>.out.* next to the inputs
//   python scripts/opencv_ref/opencv_ref_compare.py compare /tmp/lvk_cv          # per-stage report, exit code 1 on a bound violated
//
// Every call below is made the way the reference makes it; the citation is the reference's call site.
// Array files: "LVKA", int32 dtype (0 u8, 1 i32, 2 f32, 3 f64), int32 ndim, int32 dims[ndim], raw little-endian data.
#include <opencv2/calib3d.hpp>
#include <opencv2/core.hpp>
#include <opencv2/core/ocl.hpp>
#include <opencv2/features2d.hpp>
#include <opencv2/imgproc.hpp>
#include <opencv2/video/tracking.hpp>
#ifdef LVK_WITH_EIGEN
#include <Eigen/Sparse>
#include <Eigen/IterativeLinearSolvers>
#endif

#include <cstdio>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct Array { int dtype = 0; std::vector<int> dims; std::vector<uint8_t> data; };
const size_t kSize[4] = {1, 4, 4, 8};

bool load(const std::string& path, Array& a)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    char magic[4]; int32_t dtype = 0, nd = 0;
    bool ok = std::fread(magic, 1, 4, f) == 4 && std::memcmp(magic, "LVKA", 4) == 0 && std::fread(&dtype, 4, 1, f) == 1 && std::fread(&nd, 4, 1, f) == 1;
    size_t n = 1;
    a.dims.resize(ok ? nd : 0);
    for (int i = 0; ok && i < nd; i++) { int32_t d; ok = std::fread(&d, 4, 1, f) == 1; a.dims[i] = d; n *= (size_t)d; }
    a.dtype = dtype;
    if (ok) { a.data.resize(n * kSize[dtype]); ok = std::fread(a.data.data(), 1, a.data.size(), f) == a.data.size(); }
    std::fclose(f);
    return ok;
}

void save(const std::string& path, int dtype, const std::vector<int>& dims, const void* data)
{
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) { std::printf("cannot write %s\n", path.c_str()); return; }
    size_t n = 1; for (int d : dims) n *= (size_t)d;
    const int32_t dt = dtype, nd = (int32_t)dims.size();
    std::fwrite("LVKA", 1, 4, f); std::fwrite(&dt, 4, 1, f); std::fwrite(&nd, 4, 1, f);
    for (int d : dims) { const int32_t v = d; std::fwrite(&v, 4, 1, f); }
    std::fwrite(data, kSize[dtype], n, f);
    std::fclose(f);
}

cv::Mat as_mat(Array& a, int type) { return cv::Mat(a.dims[0], a.dims[1], type, a.data.data()); }
std::vector<cv::Point2f> as_points(const Array& a)
{
    const float* p = reinterpret_cast<const float*>(a.data.data());
    std::vector<cv::Point2f> v((size_t)a.dims[0]);
    for (size_t i = 0; i < v.size(); i++) v[i] = {p[2 * i], p[2 * i + 1]};
    return v;
}

} // namespace

int main(int argc, char** argv)
{
    if (argc < 2) { std::printf("usage: opencv_ref_dump <directory written by opencv_ref_compare.py export>\n"); return 2; }
    const std::string dir = std::string(argv[1]) + "/";
    // the CPU paths: the oracle restates those (the OpenCL variants are float and order-nondeterministic)
    cv::ocl::setUseOpenCL(false);
    std::printf("OpenCV %s\n", CV_VERSION);
    Array a, b, c;

    // ---- a4: cv::resize(gray, detection_resolution, INTER_AREA)                      Vision/FrameTracker.cpp:117
    for (int k = 0; load(dir + "area_" + std::to_string(k) + ".src", a); k++)
    {
        if (!load(dir + "area_" + std::to_string(k) + ".size", b)) break;
        const int* sz = reinterpret_cast<const int*>(b.data.data());                         // (width, height)
        cv::Mat dst;
        cv::resize(as_mat(a, CV_8UC1), dst, cv::Size(sz[0], sz[1]), 0, 0, cv::INTER_AREA);
        save(dir + "area_" + std::to_string(k) + ".out", 0, {dst.rows, dst.cols}, dst.data);
    }
    // ---- a3: cvtColor(BGR2GRAY)                                                      Data/VideoFrame.cpp:194
    if (load(dir + "gray.src", a))
    {
        cv::Mat bgr(a.dims[0], a.dims[1], CV_8UC3, a.data.data()), g;
        cv::cvtColor(bgr, g, cv::COLOR_BGR2GRAY);
        save(dir + "gray.out", 0, {g.rows, g.cols}, g.data);
    }
    // ---- a5: FastFeatureDetector(10, true, TYPE_9_16), setThreshold, detect(frame(bounds))   Vision/FeatureDetector.cpp:38-41,130-134
    for (int k = 0; load(dir + "fast_" + std::to_string(k) + ".img", a); k++)
    {
        if (!load(dir + "fast_" + std::to_string(k) + ".roi_thr", b)) break;                // (x, y, w, h, threshold)
        const int* r = reinterpret_cast<const int*>(b.data.data());
        auto det = cv::FastFeatureDetector::create(10, true, cv::FastFeatureDetector::TYPE_9_16);
        det->setThreshold(r[4]);
        std::vector<cv::KeyPoint> kps;
        det->detect(as_mat(a, CV_8UC1)(cv::Rect(r[0], r[1], r[2], r[3])), kps);
        std::vector<float> out;
        for (const auto& kp : kps) { out.push_back(kp.pt.x); out.push_back(kp.pt.y); out.push_back(kp.response); }
        save(dir + "fast_" + std::to_string(k) + ".out", 2, {(int)kps.size(), 3}, out.data());
    }
    // ---- a7: SparsePyrLKOpticalFlow::create({11, 11}, 3, TermCriteria(COUNT + EPS, 5, 0.01))->calc     Vision/FrameTracker.cpp:42-48,140-146
    for (int k = 0; load(dir + "lk_" + std::to_string(k) + ".prev", a); k++)
    {
        if (!load(dir + "lk_" + std::to_string(k) + ".next", b) || !load(dir + "lk_" + std::to_string(k) + ".pts", c)) break;
        auto lk = cv::SparsePyrLKOpticalFlow::create(cv::Size(11, 11), 3, cv::TermCriteria(cv::TermCriteria::COUNT + cv::TermCriteria::EPS, 5, 0.01));
        std::vector<cv::Point2f> p0 = as_points(c), p1;
        std::vector<uint8_t> status;
        lk->calc(as_mat(a, CV_8UC1), as_mat(b, CV_8UC1), p0, p1, status);
        save(dir + "lk_" + std::to_string(k) + ".out_pts", 2, {(int)p1.size(), 2}, p1.data());
        save(dir + "lk_" + std::to_string(k) + ".out_status", 0, {(int)status.size()}, status.data());
    }
    // ---- a9: findHomography(UsacParams) / estimateAffinePartial2D                    Vision/FrameTracker.cpp:337-371
    for (int k = 0; load(dir + "motion_" + std::to_string(k) + ".p1", a); k++)
    {
        if (!load(dir + "motion_" + std::to_string(k) + ".p2", b) || !load(dir + "motion_" + std::to_string(k) + ".thr", c)) break;
        const float thr = *reinterpret_cast<const float*>(c.data.data());
        cv::UsacParams params;
        params.threshold = thr; params.confidence = 0.99; params.maxIterations = 50;
        params.sampler = cv::SAMPLING_UNIFORM; params.score = cv::SCORE_METHOD_MAGSAC;
        params.loMethod = cv::LOCAL_OPTIM_SIGMA; params.loIterations = 10; params.loSampleSize = 20;
        params.final_polisher = cv::MAGSAC; params.final_polisher_iterations = 0;
        const std::vector<cv::Point2f> p1 = as_points(a), p2 = as_points(b);
        std::vector<uint8_t> mask;
        cv::Mat H = cv::findHomography(p1, p2, mask, params);
        if (H.empty()) H = cv::Mat::zeros(3, 3, CV_64F);
        H.convertTo(H, CV_64F);
        save(dir + "motion_" + std::to_string(k) + ".out_H", 3, {3, 3}, H.data);
        save(dir + "motion_" + std::to_string(k) + ".out_mask", 0, {(int)mask.size()}, mask.data());
        std::vector<uint8_t> amask;
        cv::Mat A = cv::estimateAffinePartial2D(p1, p2, amask, cv::RANSAC, thr, 50);
        if (A.empty()) A = cv::Mat::zeros(2, 3, CV_64F);
        A.convertTo(A, CV_64F);
        save(dir + "motion_" + std::to_string(k) + ".out_A", 3, {2, 3}, A.data);
        save(dir + "motion_" + std::to_string(k) + ".out_amask", 0, {(int)amask.size()}, amask.data());
    }
    // ---- a12: getGaussianKernel(2 N + 1, sigma, CV_32F)                              Vision/PathSmoother.cpp:94-98
    if (load(dir + "gauss.n_sigma", a))
    {
        const double* v = reinterpret_cast<const double*>(a.data.data());
        for (int k = 0; k < a.dims[0]; k++)
        {
            cv::Mat g = cv::getGaussianKernel((int)v[2 * k], v[2 * k + 1], CV_32F);
            save(dir + "gauss_" + std::to_string(k) + ".out", 2, {g.rows}, g.data);
        }
    }
    // ---- a14: getPerspectiveTransform(dst corners, src corners)                      Math/WarpMesh.cpp:199-214
    if (load(dir + "persp.quads", a))
    {
        const float* q = reinterpret_cast<const float*>(a.data.data());
        std::vector<double> out;
        for (int k = 0; k < a.dims[0]; k++)
        {
            cv::Point2f s[4], d[4];
            for (int i = 0; i < 4; i++) { s[i] = {q[16 * k + 2 * i], q[16 * k + 2 * i + 1]}; d[i] = {q[16 * k + 8 + 2 * i], q[16 * k + 8 + 2 * i + 1]}; }
            cv::Mat M = cv::getPerspectiveTransform(s, d);
            out.insert(out.end(), M.ptr<double>(0), M.ptr<double>(0) + 9);
        }
        save(dir + "persp.out", 3, {a.dims[0], 9}, out.data());
    }
    // ---- a14: cv::resize(offsets CV_32FC2 -> frame size, INTER_LINEAR_EXACT)         Math/WarpMesh.cpp:190
    if (load(dir + "meshmap.mesh", a) && load(dir + "meshmap.size", b))
    {
        const int* sz = reinterpret_cast<const int*>(b.data.data());
        cv::Mat mesh(a.dims[0], a.dims[1], CV_32FC2, a.data.data()), map;
        cv::resize(mesh, map, cv::Size(sz[0], sz[1]), 0, 0, cv::INTER_LINEAR_EXACT);
        save(dir + "meshmap.out", 2, {map.rows, map.cols, 2}, map.data);
    }
    // ---- f2: the plugin's chroma planes: resize(.., 2x, INTER_LINEAR) up, resize(.., 0.5, INTER_AREA) down   Interop/FrameIngest.cpp:494-557
    if (load(dir + "chroma.plane", a))
    {
        cv::Mat up, down;
        cv::resize(as_mat(a, CV_8UC1), up, cv::Size(), 2.0, 2.0, cv::INTER_LINEAR);
        cv::resize(up, down, cv::Size(), 0.5, 0.5, cv::INTER_AREA);
        save(dir + "chroma.out_up", 0, {up.rows, up.cols}, up.data);
        save(dir + "chroma.out_down", 0, {down.rows, down.cols}, down.data);
    }
#ifdef LVK_WITH_EIGEN
    // ---- a10: Eigen::LeastSquaresConjugateGradient<SparseMatrix<float>>::solveWithGuess         Vision/FrameTracker.cpp:219-222,270-276
    // (the system as the reference assembles it: float triplets -> setFromTriplets, float right-hand side, m_OptimizedMesh as the guess)
    for (int k = 0; load(dir + "lscg_" + std::to_string(k) + ".rows", a); k++)
    {
        Array cols_, vals_, rhs_, x0_, shape_;
        const std::string stem = dir + "lscg_" + std::to_string(k);
        if (!load(stem + ".cols", cols_) || !load(stem + ".vals", vals_) || !load(stem + ".b", rhs_) || !load(stem + ".x0", x0_) || !load(stem + ".shape", shape_)) break;
        const int* rr = reinterpret_cast<const int*>(a.data.data()); const int* cc = reinterpret_cast<const int*>(cols_.data.data());
        const float* vv = reinterpret_cast<const float*>(vals_.data.data()); const int* shp = reinterpret_cast<const int*>(shape_.data.data());
        std::vector<Eigen::Triplet<float>> triplets;
        for (int i = 0; i < a.dims[0]; i++) triplets.emplace_back(rr[i], cc[i], vv[i]);
        Eigen::SparseMatrix<float> A(shp[0], shp[1]);
        A.setFromTriplets(triplets.begin(), triplets.end());
        const Eigen::Map<const Eigen::VectorXf> rhs(reinterpret_cast<const float*>(rhs_.data.data()), shp[0]);
        const Eigen::Map<const Eigen::VectorXf> guess(reinterpret_cast<const float*>(x0_.data.data()), shp[1]);
        Eigen::LeastSquaresConjugateGradient<Eigen::SparseMatrix<float>> solver(A);
        const Eigen::VectorXf x = solver.solveWithGuess(rhs, guess);
        const int32_t iters = (int32_t)solver.iterations();
        save(stem + ".out", 2, {shp[1]}, x.data());
        save(stem + ".out_iters", 1, {1}, &iters);
    }
    std::printf("Eigen %d.%d.%d\n", EIGEN_WORLD_VERSION, EIGEN_MAJOR_VERSION, EIGEN_MINOR_VERSION);
#endif
    std::printf("done\n");
    return 0;
}
