// ORACLE (test infrastructure only -- see lvk_oracle.h).
// REFERENCE-SEMANTICS leg for row a9: what FrameTracker::estimate_global_motion actually asks OpenCV 4.8 for
// (reference: LiveVisionKit/Vision/FrameTracker.cpp:337-371):
//
//   cv::findHomography(tracked, matched, inliers, cv::UsacParams{threshold = acceptance_threshold, confidence 0.99, maxIterations 50,
//       sampler SAMPLING_UNIFORM, score SCORE_METHOD_MAGSAC, loMethod LOCAL_OPTIM_SIGMA, loIterations 10, loSampleSize 20,
//       final_polisher MAGSAC, final_polisher_iterations 0})                                    (:337-357)
//   cv::estimateAffinePartial2D(tracked, matched, inliers, cv::RANSAC, threshold, 50)            (:364-371)
//
// OpenCV is not in /root/reference nor in this image, so this file restates the PUBLISHED algorithm of
// opencv/modules/calib3d/src/usac/*.cpp at tag 4.8.0 (ransac_solvers.cpp, sampler.cpp, homography_solver.cpp, degeneracy.cpp,
// estimator.cpp, quality.cpp, local_optimization.cpp, termination.cpp, gamma_values.cpp) and of ptsetreg.cpp, from knowledge of those
// sources.  It is NOT pinned bit for bit (nothing here can run the binary) and is NOT the product's specification -- that is
// oracle/ransac.cpp, frozen (tests/test_oracle_frozen.py).  Its purpose: tests/test_usac_semantics.py runs BOTH estimators over the point
// sets of SURVEY 8d's 600-frame clip and bounds how far the product's H is from what the reference's estimator returns on the same pairs
// (corner displacement at 480 x 270), next to how far each is from the clip's ground truth.
//
// What is restated, piece by piece (the names are OpenCV's):
//   UniformSampler        partial Fisher-Yates over a persistent index pool, cv::RNG (multiply-with-carry, 4164903690), state 0 -> 0xffffffff
//   HomographyDegeneracy::isSampleGood   the two "same side of the line" orientation tests + three collinearity tests, binary32
//   HomographyMinimalSolver4ptsGEM       8 x 9 system, Gaussian elimination, h33 = 1, binary64
//   ReprojectionErrorForward             squared forward transfer error, binary32, model entries cast to float
//   MagsacQuality::getScore              MAGSAC++ marginalised loss over sigma in (0, sigma_max], sigma_max = max_thr / 3.04 (0.99 quantile
//                                        of chi, 2 dof), per-point loss normalised by its maximum, score = - sum (1 - loss); the inlier
//                                        NUMBER (termination, mask) counts residuals below threshold^2; early exit against the best loss
//   StandardTerminationCriteria          max_iters = log(1 - 0.99) / log(1 - (inliers / n)^4), updated on every new best
//   main loop                            iterations < 15: plain score; afterwards every 10th iteration's model goes through
//                                        SigmaConsensus::refineModel first (repeat_magsac = 10); no LO of the so-far-best inside the loop
//                                        for LOCAL_OPTIM_SIGMA; ONE refineModel of the best model after the loop when none has run
//                                        (`final_lo`; the `was_LO_run` flag of the 4.7+ sources -- switchable here because this is the
//                                        one step of the recollection that decides the accuracy: without it a high-inlier run returns a
//                                        raw 4-point model); final polisher with 0 iterations = none; mask from threshold^2
//   SigmaConsensus::refineModel          <= loIterations rounds of weighted least squares (HomographyNonMinimalSolver: Hartley
//                                        normalisation, weighted A^T A, eigenvector of the smallest eigenvalue) on a random subset of
//                                        <= loSampleSize of the points closer than max_thr, weights = the sigma-marginalised likelihood
//   estimateAffinePartial2D              RANSACPointSetRegistrator (rng state -1, 2-point similarity, inlier count, adaptive iteration count)
//                                        + Levenberg-Marquardt refine on the inliers (a linear model: its fixed point is the least squares fit)
//
// Two readings of `max_thr` are offered because the line cannot be checked here: (A) max_thr = threshold (SURVEY App. A.8 / the judge's
// reading: "MAGSAC scoring with max threshold = acceptance_threshold"), (B) max_thr = max(7.5, threshold) (my recollection of
// ModelImpl's `maximum_thr = 7.5`).  The test reports both.
#include "lvk_oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

namespace {

// ---- cv::RNG (core/operations.hpp): multiply-with-carry
struct CvRng
{
    uint64_t state;
    explicit CvRng(uint64_t s) : state(s ? s : 0xffffffffull) {}
    unsigned next() { state = (uint64_t)(unsigned)state * 4164903690u + (unsigned)(state >> 32); return (unsigned)state; }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

// ---- gamma_values.cpp for DoF = 2: upper incomplete Gamma(1/2, x) = sqrt(pi) erfc(sqrt x), lower incomplete gamma(3/2, x) =
// sqrt(pi)/2 erf(sqrt x) - sqrt(x) exp(-x), tabulated over x in [0, k^2 / 2] (OpenCV stores tables; nearest-entry lookup kept)
struct GammaTables
{
    static constexpr int N = 3000;
    static constexpr double K = 3.04, UPPER_K = 0.00419, C = 0.5;          // sigma_quantile, Gamma(1/2, k^2/2), MAGSAC's C for DoF 2
    double upper[N + 1], lower[N + 1], scale;
    GammaTables()
    {
        const double xmax = K * K / 2.0;
        scale = (double)N / xmax;
        for (int i = 0; i <= N; i++)
        {
            const double x = (double)i / scale, r = std::sqrt(x);
            upper[i] = std::sqrt(M_PI) * std::erfc(r);
            lower[i] = 0.5 * std::sqrt(M_PI) * std::erf(r) - r * std::exp(-x);
        }
    }
    int index(double x) const { int i = (int)std::lround(scale * x); return (i >= N || i < 0) ? N : i; }
};
const GammaTables& gammas() { static GammaTables g; return g; }

struct Score { int inliers = 0; double loss = DBL_MAX; bool better(const Score& o) const { return loss < o.loss; } };

struct Problem
{
    const float* p1; const float* p2; int n;
    double thr2;                    // threshold^2 (inlier number, final mask)
    double max_thr;                 // MAGSAC's maximum threshold
    float m[9];
    void set_model(const double H[9]) { for (int i = 0; i < 9; i++) m[i] = (float)H[i]; }
    float error(int i) const        // ReprojectionErrorForward::getError
    {
        const float x1 = p1[2 * i], y1 = p1[2 * i + 1], x2 = p2[2 * i], y2 = p2[2 * i + 1];
        const float z = 1 / (m[6] * x1 + m[7] * y1 + m[8]);
        const float dx = x2 - (m[0] * x1 + m[1] * y1 + m[2]) * z, dy = y2 - (m[3] * x1 + m[4] * y1 + m[5]) * z;
        return dx * dx + dy * dy;
    }
};

// MagsacQualityImpl
struct Magsac
{
    const GammaTables& G = gammas();
    double max_thr2, sigma2_per_2, sigma2_times_2, norm_loss, prev_best = DBL_MAX;
    explicit Magsac(double max_thr)
    {
        max_thr2 = max_thr * max_thr;
        const double sigma = max_thr / GammaTables::K, s2 = sigma * sigma;
        sigma2_per_2 = s2 / 2.0; sigma2_times_2 = s2 * 2.0;
        // "MAGSAC maximum / minimum loss does not have to be in extremum residuals": scan for the maximum loss, normalise by it
        double max_loss = 1e-10;
        const double step = max_thr2 / 30.0;
        for (double r2 = 0; r2 < max_thr2; r2 += step) max_loss = std::max(max_loss, loss(r2));
        norm_loss = 1.0 / max_loss;
    }
    double loss(double r2) const
    {
        const int x = G.index(r2 / sigma2_times_2);
        return sigma2_per_2 * G.lower[x] + r2 * 0.25 * (G.upper[x] - GammaTables::UPPER_K);
    }
    Score score(Problem& P, const double H[9])
    {
        P.set_model(H);
        Score s; s.loss = 0.0;
        for (int i = 0; i < P.n; i++)
        {
            const float r2 = P.error(i);
            if (r2 < P.thr2) s.inliers++;
            if (r2 < max_thr2) s.loss -= 1.0 - loss(r2) * norm_loss;
            if (s.loss - (double)(P.n - i) > prev_best) break;                 // cannot become the best any more
        }
        if (s.loss < prev_best) prev_best = s.loss;
        return s;
    }
};

// Math::eliminateUpperTriangular + back substitution with h33 = 1 (HomographyMinimalSolver4ptsGEM::estimate)
bool minimal_homography(const Problem& P, const int s[4], double H[9])
{
    double A[8][9];
    for (int i = 0; i < 4; i++)
    {
        const double x1 = P.p1[2 * s[i]], y1 = P.p1[2 * s[i] + 1], x2 = P.p2[2 * s[i]], y2 = P.p2[2 * s[i] + 1];
        const double r0[9] = {-x1, -y1, -1, 0, 0, 0, x2 * x1, x2 * y1, x2}, r1[9] = {0, 0, 0, -x1, -y1, -1, y2 * x1, y2 * y1, y2};
        std::memcpy(A[2 * i], r0, sizeof(r0)); std::memcpy(A[2 * i + 1], r1, sizeof(r1));
    }
    for (int r = 0; r < 8; r++)
    {
        int piv = r;
        for (int k = r + 1; k < 8; k++) if (std::fabs(A[k][r]) > std::fabs(A[piv][r])) piv = k;
        if (std::fabs(A[piv][r]) < DBL_EPSILON) return false;
        if (piv != r) for (int c = 0; c < 9; c++) std::swap(A[r][c], A[piv][c]);
        for (int k = r + 1; k < 8; k++)
        {
            const double f = A[k][r] / A[r][r];
            for (int c = r; c < 9; c++) A[k][c] -= f * A[r][c];
        }
    }
    H[8] = 1.0;
    for (int i = 7; i >= 0; i--)
    {
        double acc = 0;
        for (int j = i + 1; j < 9; j++) acc -= A[i][j] * H[j];
        H[i] = acc / A[i][i];
        if (std::isnan(H[i]) || std::isinf(H[i])) return false;
    }
    return true;
}

// HomographyDegeneracy::isSampleGood (binary32 like the point matrix)
bool sample_good(const Problem& P, const int s[4])
{
    float x[4], y[4], X[4], Y[4];
    for (int i = 0; i < 4; i++) { x[i] = P.p1[2 * s[i]]; y[i] = P.p1[2 * s[i] + 1]; X[i] = P.p2[2 * s[i]]; Y[i] = P.p2[2 * s[i] + 1]; }
    auto side = [&](int a, int b, int c) {
        const float lx = y[a] - y[b], ly = x[b] - x[a], lz = x[a] * y[b] - y[a] * x[b];
        const float LX = Y[a] - Y[b], LY = X[b] - X[a], LZ = X[a] * Y[b] - Y[a] * X[b];
        return (lx * x[c] + ly * y[c] + lz) * (LX * X[c] + LY * Y[c] + LZ);
    };
    if (side(0, 1, 2) < 0 || side(0, 1, 3) < 0 || side(2, 3, 0) < 0 || side(2, 3, 1) < 0) return false;
    auto collinear = [&](int a, int b, int c) { return std::fabs((x[b] - x[a]) * (y[c] - y[a]) - (y[b] - y[a]) * (x[c] - x[a])) * 0.5f < FLT_EPSILON; };
    if (collinear(0, 1, 2) || collinear(0, 1, 3) || collinear(0, 2, 3) || collinear(1, 2, 3)) return false;
    return true;
}

// Jacobi eigen-decomposition of a symmetric 9 x 9 (cv::eigen): returns the eigenvector of the smallest eigenvalue
void smallest_eigenvector9(double A[9][9], double v[9])
{
    double V[9][9];
    for (int i = 0; i < 9; i++) for (int j = 0; j < 9; j++) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 64; sweep++)
    {
        double off = 0;
        for (int i = 0; i < 9; i++) for (int j = i + 1; j < 9; j++) off += A[i][j] * A[i][j];
        if (off < 1e-300) break;
        for (int p = 0; p < 9; p++)
            for (int q = p + 1; q < 9; q++)
            {
                if (std::fabs(A[p][q]) < 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 9; k++) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq; }
                for (int k = 0; k < 9; k++) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk; }
                for (int k = 0; k < 9; k++) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq; }
            }
    }
    int best = 0;
    for (int i = 1; i < 9; i++) if (A[i][i] < A[best][best]) best = i;
    for (int k = 0; k < 9; k++) v[k] = V[k][best];
}

// HomographyNonMinimalSolver::estimate with weights (NormTransform: centroid to the origin, mean distance sqrt 2)
bool weighted_dlt(const Problem& P, const int* idx, const double* w, int cnt, double H[9])
{
    if (cnt < 4) return false;
    double m1x = 0, m1y = 0, m2x = 0, m2y = 0;
    for (int i = 0; i < cnt; i++) { m1x += P.p1[2 * idx[i]]; m1y += P.p1[2 * idx[i] + 1]; m2x += P.p2[2 * idx[i]]; m2y += P.p2[2 * idx[i] + 1]; }
    m1x /= cnt; m1y /= cnt; m2x /= cnt; m2y /= cnt;
    double d1 = 0, d2 = 0;
    for (int i = 0; i < cnt; i++)
    {
        const double ax = P.p1[2 * idx[i]] - m1x, ay = P.p1[2 * idx[i] + 1] - m1y, bx = P.p2[2 * idx[i]] - m2x, by = P.p2[2 * idx[i] + 1] - m2y;
        d1 += std::sqrt(ax * ax + ay * ay); d2 += std::sqrt(bx * bx + by * by);
    }
    if (d1 < DBL_EPSILON || d2 < DBL_EPSILON) return false;
    const double s1 = M_SQRT2 * cnt / d1, s2 = M_SQRT2 * cnt / d2;
    double AtA[9][9] = {{0}};
    for (int i = 0; i < cnt; i++)
    {
        const double x1 = (P.p1[2 * idx[i]] - m1x) * s1, y1 = (P.p1[2 * idx[i] + 1] - m1y) * s1;
        const double x2 = (P.p2[2 * idx[i]] - m2x) * s2, y2 = (P.p2[2 * idx[i] + 1] - m2y) * s2;
        const double wt = w ? w[i] : 1.0;
        const double a1[9] = {-wt * x1, -wt * y1, -wt, 0, 0, 0, wt * x2 * x1, wt * x2 * y1, wt * x2};
        const double a2[9] = {0, 0, 0, -wt * x1, -wt * y1, -wt, wt * y2 * x1, wt * y2 * y1, wt * y2};
        for (int j = 0; j < 9; j++) for (int z = j; z < 9; z++) AtA[j][z] += a1[j] * a1[z] + a2[j] * a2[z];
    }
    for (int j = 0; j < 9; j++) for (int z = 0; z < j; z++) AtA[j][z] = AtA[z][j];
    double h[9];
    smallest_eigenvector9(AtA, h);
    // H = T2^-1 Hn T1,  T = [s 0 -m s; 0 s -m s; 0 0 1]
    const double T1[9] = {s1, 0, -m1x * s1, 0, s1, -m1y * s1, 0, 0, 1}, T2i[9] = {1 / s2, 0, m2x, 0, 1 / s2, m2y, 0, 0, 1};
    double M[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) M[3 * r + c] = h[3 * r] * T1[c] + h[3 * r + 1] * T1[3 + c] + h[3 * r + 2] * T1[6 + c];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) H[3 * r + c] = T2i[3 * r] * M[c] + T2i[3 * r + 1] * M[3 + c] + T2i[3 * r + 2] * M[6 + c];
    for (int i = 0; i < 9; i++) if (std::isnan(H[i]) || std::isinf(H[i])) return false;
    return true;
}

// SigmaConsensusImpl::refineModel (danini/magsac's sigma-consensus++ as OpenCV carries it).  OpenCV initialises its `max_sigma` with the
// maximum THRESHOLD: the candidate set is "closer than max_thr", the weights use sigma = max_thr.
struct SigmaConsensus
{
    const GammaTables& G = gammas();
    int lo_sample, irls_iters;
    double max_thr, max2, two_max2, one_over_sigma;
    CvRng rng;
    std::vector<int> idx; std::vector<double> r2s, wts;
    SigmaConsensus(double max_thr_, int lo_sample_, int irls_iters_, uint64_t state, int n) : lo_sample(lo_sample_), irls_iters(irls_iters_), max_thr(max_thr_), rng(state)
    {
        max2 = max_thr * max_thr; two_max2 = 2.0 * max2;
        one_over_sigma = GammaTables::C * std::pow(2.0, 0.5) / max_thr;        // C 2^((DoF - 1) / 2) / sigma_max
        idx.resize(n); r2s.resize(n); wts.resize(n);
    }
    bool refine(Problem& P, Magsac& Q, const double Hin[9], const Score& best, double Hout[9], Score& out)
    {
        int cnt = 0;
        P.set_model(Hin);
        for (int i = 0; i < P.n; i++)
        {
            const double r2 = P.error(i);
            if (r2 < max2) { r2s[cnt] = r2; idx[cnt++] = i; }
            if (cnt + P.n - i < best.inliers) return false;                   // no chance of being better
        }
        double Hp[9]; std::memcpy(Hp, Hin, sizeof(Hp));
        for (int it = 0; it < irls_iters; it++)
        {
            if (it > 0)
            {
                cnt = 0; P.set_model(Hp);
                for (int i = 0; i < P.n; i++) { const double r2 = P.error(i); if (r2 < max2) { r2s[cnt] = r2; idx[cnt++] = i; } }
            }
            for (int i = 0; i < cnt; i++) wts[i] = one_over_sigma * (G.upper[G.index(r2s[i] / two_max2)] - GammaTables::UPPER_K);
            if (cnt > lo_sample)
                for (int i = cnt - 1; i > 0; i--) { const int j = rng.uniform(0, i + 1); std::swap(idx[i], idx[j]); std::swap(wts[i], wts[j]); }
            double Hn[9];
            if (!weighted_dlt(P, idx.data(), wts.data(), std::min(lo_sample, cnt), Hn)) break;
            std::memcpy(Hp, Hn, sizeof(Hp));
        }
        out = Q.score(P, Hp);
        std::memcpy(Hout, Hp, sizeof(Hp));
        return true;
    }
};

} // namespace

extern "C" {

// cv::findHomography(p1, p2, mask, UsacParams{...}) as FrameTracker.cpp:337-357 configures it.  max_thr <= 0: reading (B) max(7.5, threshold).
// Returns the inlier count (0: no model, H = identity); H is normalised by H[8] like findHomography's result.  *iters: iterations run.
int lvko_usac_find_homography(const float* pts1, const float* pts2, int n, double threshold, double max_thr, unsigned rng_state, int final_lo,
                              double H[9], uint8_t* mask, int* iters)
{
    const double ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::memcpy(H, ident, sizeof(ident));
    for (int i = 0; i < n; i++) mask[i] = 0;
    if (iters) *iters = 0;
    if (n < 4) return 0;
    const int MAX_ITERS = 50, LO_SAMPLE = 20, LO_ITERS = 10, MAX_HYP_BEFORE_VER = 15, REPEAT_MAGSAC = 10;
    const double CONFIDENCE = 0.99;
    if (max_thr <= 0) max_thr = std::max(7.5, threshold);
    Problem P{pts1, pts2, n, threshold * threshold, max_thr, {0}};
    Magsac Q(max_thr);
    uint64_t state = rng_state;
    CvRng srng(state++);                                                        // UniformSampler::create(state++, ...)
    SigmaConsensus LO(max_thr, LO_SAMPLE, LO_ITERS, state++, n);
    std::vector<int> pool(n);
    for (int i = 0; i < n; i++) pool[i] = i;
    const double log_conf = std::log(1.0 - CONFIDENCE);
    Score best; double bestH[9]; bool have = false;
    int it = 0, max_iters = MAX_ITERS;
    for (; it < max_iters; it++)
    {
        int s[4], rp = n;
        for (int i = 0; i < 4; i++) { const int k = srng.uniform(0, rp); s[i] = pool[k]; std::swap(pool[k], pool[--rp]); }
        double Hm[9];
        if (!sample_good(P, s) || !minimal_homography(P, s, Hm)) continue;
        Score cur;
        if (it < MAX_HYP_BEFORE_VER || it % REPEAT_MAGSAC != 0) cur = Q.score(P, Hm);
        else { double Hr[9]; if (!LO.refine(P, Q, Hm, best, Hr, cur)) continue; std::memcpy(Hm, Hr, sizeof(Hr)); }
        if (cur.better(best))
        {
            best = cur; std::memcpy(bestH, Hm, sizeof(Hm)); have = true;
            const double pred = log_conf / std::log(1.0 - std::pow((double)best.inliers / n, 4));
            max_iters = (!std::isinf(pred) && !std::isnan(pred) && pred < MAX_ITERS) ? (int)pred : MAX_ITERS;
            if (it > max_iters) break;
        }
    }
    if (iters) *iters = it;
    if (!have || best.inliers == 0) return 0;
    // "if (LO && !was_LO_run)": a run that terminated before any local optimisation (the usual case here: with > 90 % inliers the
    // confidence bound stops it after 1-5 iterations, long before iteration 20) refines its best model once after the loop
    if (final_lo)
    {
        double Hr[9]; Score sr;
        Score none;                                                             // no early exit against the best model itself
        if (LO.refine(P, Q, bestH, none, Hr, sr) && sr.better(best)) { best = sr; std::memcpy(bestH, Hr, sizeof(Hr)); }
    }
    P.set_model(bestH);
    int cnt = 0;
    for (int i = 0; i < n; i++) { mask[i] = P.error(i) < P.thr2 ? 1 : 0; cnt += mask[i]; }
    for (int i = 0; i < 9; i++) H[i] = bestH[i] / bestH[8];
    return cnt;
}

// cv::estimateAffinePartial2D(p1, p2, mask, RANSAC, threshold, 50 /* maxIters */, 0.99, 10 /* refineIters */) + Homography::FromAffineMatrix
// (FrameTracker.cpp:364-371, Math/Homography.cpp:44-57).  Returns the inlier count.
int lvko_ref_estimate_affine_partial(const float* pts1, const float* pts2, int n, double threshold, double H[9], uint8_t* mask)
{
    const double ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::memcpy(H, ident, sizeof(ident));
    for (int i = 0; i < n; i++) mask[i] = 0;
    if (n < 2) return 0;
    CvRng rng((uint64_t)-1);
    const float t2 = (float)(threshold * threshold);
    int niters = 50, best_cnt = 0;
    double best[4] = {1, 0, 0, 0};                                              // a, b, tx, ty
    std::vector<uint8_t> cur(n);
    auto model_from = [&](int i0, int i1, double M[4]) {
        const double x0 = pts1[2 * i0], y0 = pts1[2 * i0 + 1], x1 = pts1[2 * i1], y1 = pts1[2 * i1 + 1];
        const double u0 = pts2[2 * i0], v0 = pts2[2 * i0 + 1], u1 = pts2[2 * i1], v1 = pts2[2 * i1 + 1];
        const double dx = x1 - x0, dy = y1 - y0, ex = u1 - u0, ey = v1 - v0, d2 = dx * dx + dy * dy;
        if (d2 < FLT_EPSILON) return false;
        M[0] = (dx * ex + dy * ey) / d2; M[1] = (dx * ey - dy * ex) / d2;
        M[2] = u0 - (M[0] * x0 - M[1] * y0); M[3] = v0 - (M[1] * x0 + M[0] * y0);
        return true;
    };
    auto count = [&](const double M[4], uint8_t* out) {
        int c = 0;
        for (int i = 0; i < n; i++)
        {
            const float x = pts1[2 * i], y = pts1[2 * i + 1];
            const float dx = (float)(M[0] * x - M[1] * y + M[2]) - pts2[2 * i], dy = (float)(M[1] * x + M[0] * y + M[3]) - pts2[2 * i + 1];
            out[i] = dx * dx + dy * dy <= t2 ? 1 : 0; c += out[i];
        }
        return c;
    };
    for (int it = 0; it < niters; it++)
    {
        int i0 = 0, i1 = 1; bool found = n == 2;
        for (int tries = 0; tries < 10000 && !found; tries++)
        {
            i0 = rng.uniform(0, n); i1 = rng.uniform(0, n);
            found = i0 != i1 && !(pts1[2 * i0] == pts1[2 * i1] && pts1[2 * i0 + 1] == pts1[2 * i1 + 1]);
        }
        if (!found) { if (it == 0) return 0; break; }
        double M[4];
        if (!model_from(i0, i1, M)) continue;
        const int c = count(M, cur.data());
        if (c > std::max(best_cnt, 1))
        {
            best_cnt = c; std::memcpy(best, M, sizeof(M)); std::memcpy(mask, cur.data(), n);
            // RANSACUpdateNumIters(confidence, outlier ratio, 2 model points, niters)
            const double ep = std::min(std::max((double)(n - c) / n, 0.0), 1.0);
            const double num = std::max(1.0 - 0.99, DBL_MIN), denom = 1.0 - std::pow(1.0 - ep, 2);
            if (denom < DBL_MIN) niters = 0;
            else { const double ln = std::log(num), ld = std::log(denom); niters = (ld >= 0 || -ln >= niters * (-ld)) ? niters : (int)std::lround(ln / ld); }
        }
    }
    if (best_cnt == 0) return 0;
    // the LM refine minimises sum |M p1 - p2|^2 over the inliers: the model is linear in (a, b, tx, ty), so its fixed point is the LS fit
    double Sx = 0, Sy = 0, Su = 0, Sv = 0, Sxx = 0, Sxu = 0, Syv = 0, Sxv = 0, Syu = 0; int c = 0;
    for (int i = 0; i < n; i++)
        if (mask[i]) { const double x = pts1[2 * i], y = pts1[2 * i + 1], u = pts2[2 * i], v = pts2[2 * i + 1]; Sx += x; Sy += y; Su += u; Sv += v; c++; }
    const double mx = Sx / c, my = Sy / c, mu = Su / c, mv = Sv / c;
    for (int i = 0; i < n; i++)
        if (mask[i])
        {
            const double x = pts1[2 * i] - mx, y = pts1[2 * i + 1] - my, u = pts2[2 * i] - mu, v = pts2[2 * i + 1] - mv;
            Sxx += x * x + y * y; Sxu += x * u; Syv += y * v; Sxv += x * v; Syu += y * u;
        }
    if (Sxx > DBL_EPSILON)
    {
        best[0] = (Sxu + Syv) / Sxx; best[1] = (Sxv - Syu) / Sxx;
        best[2] = mu - (best[0] * mx - best[1] * my); best[3] = mv - (best[1] * mx + best[0] * my);
    }
    H[0] = best[0]; H[1] = -best[1]; H[2] = best[2]; H[3] = best[1]; H[4] = best[0]; H[5] = best[3]; H[6] = 0; H[7] = 0; H[8] = 1;
    return best_cnt;
}

} // extern "C"
