// ORACLE (test infrastructure only -- see lvk_oracle.h).
// Robust global motion estimate: stands in for cv::findHomography(UsacParams) and cv::estimateAffinePartial2D
// as called by FrameTracker::estimate_global_motion (reference: Vision/FrameTracker.cpp:325-375).
//
// NOT a restatement of OpenCV's USAC (a large randomised framework whose source is not in /root/reference and
// whose sampling is not reproducible off its own RNG): SURVEY.md Appendix A.8 fixes OUR specification instead --
// same inputs, same outputs (3x3 double normalised by H22 / 2x3 similarity, 0/1 inlier mask), same threshold
// semantics (reprojection error against `acceptance_threshold`), deterministic and GPU-friendly:
//
//   1. K = 128 hypotheses.  Hypothesis h draws its minimal sample (4 pairs / 2 pairs, distinct indices) from a
//      counter-based SplitMix64 stream seeded by h only (like USAC's fixed RNG state 0, the schedule depends on
//      nothing but n).
//   2. Model from the minimal sample: 8x8 linear system (as cv::getPerspectiveTransform) / closed-form similarity.
//   3. Score = sum over all pairs of  floor(1024 * max(0, 1 - e^2 / t^2))  with e = forward reprojection error --
//      an MSAC/MAGSAC-style truncated quadratic, accumulated as an exact integer.  Best score wins, ties go to
//      the lower hypothesis index.
//   4. Local optimisation: up to 3 rounds of least squares on the current inliers (e^2 <= t^2), accepted while
//      the score strictly improves.  The normal-equation sums are taken in "block order" (256 strided partial sums
//      combined by a binary tree) so that a 256-thread GPU block reproduces them bit for bit.
//   All arithmetic is binary64, separately rounded (no contraction).
#include "lvk_oracle.h"

#include <cmath>
#include <cstring>
#include <vector>
#include <algorithm>

namespace {

constexpr int K_HYPOTHESES = 128;
constexpr int LO_ROUNDS = 3;

inline uint64_t splitmix64(uint64_t& s)
{
    s += 0x9E3779B97F4A7C15ull;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Draw `m` distinct indices in [0, n) for hypothesis h.  Returns false if 32 draws did not produce them.
bool draw_sample(int h, int n, int m, int* idx)
{
    uint64_t s = 0x4C564B31ull ^ ((uint64_t)(h + 1) * 0xD1B54A32D192ED03ull);
    int got = 0;
    for (int draw = 0; draw < 32 && got < m; draw++)
    {
        const int c = (int)((splitmix64(s) >> 32) % (uint64_t)n);
        bool dup = false;
        for (int j = 0; j < got; j++) dup = dup || (idx[j] == c);
        if (!dup) idx[got++] = c;
    }
    return got == m;
}

// Gaussian elimination with partial pivoting, n <= 8.  A is n x n row major, overwritten.  Pivots enter as reciprocals, in the
// elimination AND in the back substitution (x_i = s_i * (1 / a_ii): the reciprocals do not depend on the solution, so the sixteen
// dependent divisions of the textbook form -- what a GPU wavefront spends most of this solve waiting for -- become eight).
bool solve_n(double* A, double* b, int n)
{
    double rcp[8];
    for (int i = 0; i < n; i++)
    {
        int piv = i;
        for (int j = i + 1; j < n; j++) if (std::fabs(A[j * n + i]) > std::fabs(A[piv * n + i])) piv = j;
        if (std::fabs(A[piv * n + i]) < 1e-10) return false;
        if (piv != i) { for (int q = 0; q < n; q++) std::swap(A[i * n + q], A[piv * n + q]); std::swap(b[i], b[piv]); }
        const double inv = 1.0 / A[i * n + i];
        rcp[i] = inv;
        for (int j = i + 1; j < n; j++)
        {
            const double f = A[j * n + i] * inv;
            for (int q = i + 1; q < n; q++) A[j * n + q] = A[j * n + q] - f * A[i * n + q];
            b[j] = b[j] - f * b[i];
        }
    }
    for (int i = n - 1; i >= 0; i--)
    {
        double s = b[i];
        for (int q = i + 1; q < n; q++) s = s - A[i * n + q] * b[q];
        b[i] = s * rcp[i];
    }
    return true;
}

bool homography_from_4(const float* p1, const float* p2, const int* idx, double H[9])
{
    double A[64], b[8];
    for (int i = 0; i < 4; i++)
    {
        const double x = p1[2 * idx[i]], y = p1[2 * idx[i] + 1], u = p2[2 * idx[i]], v = p2[2 * idx[i] + 1];
        double* r0 = A + (size_t)i * 8; double* r1 = A + (size_t)(i + 4) * 8;
        r0[0] = x; r0[1] = y; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -x * u; r0[7] = -y * u; b[i] = u;
        r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = x; r1[4] = y; r1[5] = 1; r1[6] = -x * v; r1[7] = -y * v; b[i + 4] = v;
    }
    if (!solve_n(A, b, 8)) return false;
    for (int q = 0; q < 8; q++) H[q] = b[q];
    H[8] = 1.0;
    return true;
}

bool similarity_from_2(const float* p1, const float* p2, const int* idx, double H[9])
{
    const double x0 = p1[2 * idx[0]], y0 = p1[2 * idx[0] + 1], x1 = p1[2 * idx[1]], y1 = p1[2 * idx[1] + 1];
    const double u0 = p2[2 * idx[0]], v0 = p2[2 * idx[0] + 1], u1 = p2[2 * idx[1]], v1 = p2[2 * idx[1] + 1];
    const double dx = x1 - x0, dy = y1 - y0, ex = u1 - u0, ey = v1 - v0;
    const double d2 = dx * dx + dy * dy;
    if (d2 < 1e-10) return false;
    const double a = (dx * ex + dy * ey) / d2, b = (dx * ey - dy * ex) / d2;
    H[0] = a; H[1] = -b; H[2] = u0 - (a * x0 - b * y0);
    H[3] = b; H[4] = a;  H[5] = v0 - (b * x0 + a * y0);
    H[6] = 0; H[7] = 0;  H[8] = 1;
    return true;
}

inline double reproj_err2(const double H[9], double x, double y, double u, double v)
{
    const double w = H[6] * x + H[7] * y + H[8];
    if (std::fabs(w) < 1e-12) return 1e300;
    const double px = (H[0] * x + H[1] * y + H[2]) / w, py = (H[3] * x + H[4] * y + H[5]) / w;
    const double ex = px - u, ey = py - v;
    return ex * ex + ey * ey;
}

long long score_model(const double H[9], const float* p1, const float* p2, int n, double t2, uint8_t* mask, int* ninl)
{
    long long score = 0;
    int cnt = 0;
    for (int i = 0; i < n; i++)
    {
        const double e2 = reproj_err2(H, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1]);
        const bool in = e2 <= t2;
        if (in) { score += (long long)((1.0 - e2 / t2) * 1024.0); cnt++; }
        if (mask) mask[i] = in ? 1 : 0;
    }
    if (ninl) *ninl = cnt;
    return score;
}

// "Block order" sum: the term of pair i goes to partial (i mod 256) -- chosen by the POINT index, terms of non-inliers
// are skipped, each partial accumulates in increasing i -- and the 256 partials are combined by the binary tree
// v[j] += v[j + o] for j < o, o = 128, 64, ..., 1.  A 256-thread GPU block reproduces exactly this order.
struct WaveAcc
{
    static constexpr int P = 256;
    double part[P];
    WaveAcc() { for (double& p : part) p = 0.0; }
    void add(int i, double v) { part[i % P] = part[i % P] + v; }
    double total() const
    {
        double v[P]; std::memcpy(v, part, sizeof(v));
        for (int o = P / 2; o >= 1; o >>= 1)
            for (int l = 0; l < o; l++) v[l] = v[l] + v[l + o];
        return v[0];
    }
};

bool refit_homography(const float* p1, const float* p2, int n, const uint8_t* mask, double cx, double cy, double sc, double H[9])
{
    // normal equations of the inhomogeneous DLT (h22 = 1) on normalised coordinates
    WaveAcc N[36], g[8];
    for (int i = 0; i < n; i++)
    {
        if (!mask[i]) continue;
        const double x = (p1[2 * i] - cx) * sc, y = (p1[2 * i + 1] - cy) * sc;
        const double u = (p2[2 * i] - cx) * sc, v = (p2[2 * i + 1] - cy) * sc;
        const double r0[8] = {x, y, 1, 0, 0, 0, -x * u, -y * u};
        const double r1[8] = {0, 0, 0, x, y, 1, -x * v, -y * v};
        int k = 0;
        for (int a = 0; a < 8; a++)
            for (int b = a; b < 8; b++, k++)
                N[k].add(i, r0[a] * r0[b] + r1[a] * r1[b]);
        for (int a = 0; a < 8; a++) g[a].add(i, r0[a] * u + r1[a] * v);
    }
    double A[64], b[8];
    int k = 0;
    for (int a = 0; a < 8; a++)
        for (int c = a; c < 8; c++, k++) { const double t = N[k].total(); A[a * 8 + c] = t; A[c * 8 + a] = t; }
    for (int a = 0; a < 8; a++) b[a] = g[a].total();
    if (!solve_n(A, b, 8)) return false;
    // denormalise: H = T^-1 * Hn * T with T = [sc 0 -cx*sc; 0 sc -cy*sc; 0 0 1]
    const double Hn[9] = {b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], 1.0};
    const double T[9] = {sc, 0, -cx * sc, 0, sc, -cy * sc, 0, 0, 1};
    const double Ti[9] = {1.0 / sc, 0, cx, 0, 1.0 / sc, cy, 0, 0, 1};
    double M[9], R[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) M[r * 3 + c] = (Hn[r * 3] * T[c] + Hn[r * 3 + 1] * T[3 + c]) + Hn[r * 3 + 2] * T[6 + c];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R[r * 3 + c] = (Ti[r * 3] * M[c] + Ti[r * 3 + 1] * M[3 + c]) + Ti[r * 3 + 2] * M[6 + c];
    if (std::fabs(R[8]) < 1e-12) return false;
    for (int q = 0; q < 9; q++) H[q] = R[q] / R[8];
    return true;
}

bool refit_similarity(const float* p1, const float* p2, int n, const uint8_t* mask, double cx, double cy, double sc, double H[9])
{
    // unknowns (a, b, tx, ty): u = a x - b y + tx, v = b x + a y + ty  -> 4x4 normal equations
    WaveAcc N[10], g[4];
    for (int i = 0; i < n; i++)
    {
        if (!mask[i]) continue;
        const double x = (p1[2 * i] - cx) * sc, y = (p1[2 * i + 1] - cy) * sc;
        const double u = (p2[2 * i] - cx) * sc, v = (p2[2 * i + 1] - cy) * sc;
        const double r0[4] = {x, -y, 1, 0};
        const double r1[4] = {y, x, 0, 1};
        int k = 0;
        for (int a = 0; a < 4; a++)
            for (int b = a; b < 4; b++, k++)
                N[k].add(i, r0[a] * r0[b] + r1[a] * r1[b]);
        for (int a = 0; a < 4; a++) g[a].add(i, r0[a] * u + r1[a] * v);
    }
    double A[16], b[4];
    int k = 0;
    for (int a = 0; a < 4; a++)
        for (int c = a; c < 4; c++, k++) { const double t = N[k].total(); A[a * 4 + c] = t; A[c * 4 + a] = t; }
    for (int a = 0; a < 4; a++) b[a] = g[a].total();
    if (!solve_n(A, b, 4)) return false;
    // denormalise: p' = R p + t in normalised coords  ->  original coords
    const double a = b[0], bb = b[1], tx = b[2], ty = b[3];
    H[0] = a; H[1] = -bb; H[2] = (tx / sc + cx) - (a * cx - bb * cy);
    H[3] = bb; H[4] = a;  H[5] = (ty / sc + cy) - (bb * cx + a * cy);
    H[6] = 0; H[7] = 0; H[8] = 1;
    return true;
}

int estimate(const float* p1, const float* p2, int n, double threshold, double region_w, double region_h,
             bool full_homography, double H[9], uint8_t* mask)
{
    const double ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::memcpy(H, ident, sizeof(ident));
    for (int i = 0; i < n; i++) mask[i] = 0;
    const int m = full_homography ? 4 : 2;
    if (n < m) return -1;
    const double t2 = threshold * threshold;
    long long best_score = -1; int best_h = -1; double best_H[9];
    for (int h = 0; h < K_HYPOTHESES; h++)
    {
        int idx[4]; double Hh[9];
        if (!draw_sample(h, n, m, idx)) continue;
        if (!(full_homography ? homography_from_4(p1, p2, idx, Hh) : similarity_from_2(p1, p2, idx, Hh))) continue;
        const long long s = score_model(Hh, p1, p2, n, t2, nullptr, nullptr);
        if (s > best_score) { best_score = s; best_h = h; std::memcpy(best_H, Hh, sizeof(Hh)); }
    }
    if (best_h < 0) return -2;
    std::vector<uint8_t> cur(n), trial(n);
    int ninl = 0;
    best_score = score_model(best_H, p1, p2, n, t2, cur.data(), &ninl);
    const double cx = region_w * 0.5, cy = region_h * 0.5, sc = 2.0 / (region_w + region_h);
    for (int round = 0; round < LO_ROUNDS; round++)
    {
        if (ninl < m) break;
        double Hr[9];
        const bool ok = full_homography ? refit_homography(p1, p2, n, cur.data(), cx, cy, sc, Hr)
                                        : refit_similarity(p1, p2, n, cur.data(), cx, cy, sc, Hr);
        if (!ok) break;
        int nt = 0;
        const long long s = score_model(Hr, p1, p2, n, t2, trial.data(), &nt);
        if (s <= best_score) break;
        best_score = s; ninl = nt; std::memcpy(best_H, Hr, sizeof(Hr)); cur.swap(trial);
    }
    std::memcpy(H, best_H, sizeof(best_H));
    std::memcpy(mask, cur.data(), n);
    return ninl;
}

} // namespace

extern "C" {

// Stands in for cv::findHomography(tracked, matched, mask, UsacParams{threshold,...}) (FrameTracker.cpp:337-357).
// (region_w, region_h) = tracking resolution (used only to condition the least-squares refit).  Returns #inliers or < 0.
int lvko_find_homography(const float* pts1, const float* pts2, int n, double threshold, double region_w, double region_h,
                         double H[9], uint8_t* mask)
{
    return estimate(pts1, pts2, n, threshold, region_w, region_h, true, H, mask);
}

// Stands in for cv::estimateAffinePartial2D(..., RANSAC, threshold, 50) + Homography::FromAffineMatrix
// (FrameTracker.cpp:359-374, Math/Homography.cpp:44-57): 4-dof similarity embedded in a 3x3.
int lvko_estimate_affine_partial(const float* pts1, const float* pts2, int n, double threshold, double region_w, double region_h,
                                 double H[9], uint8_t* mask)
{
    return estimate(pts1, pts2, n, threshold, region_w, region_h, false, H, mask);
}

} // extern "C"
