// ORACLE (test infrastructure only -- see lvk_oracle.h).
// CPU restatement of FrameTracker::estimate_local_motions + generate_mesh_constraints
// (reference: Vision/FrameTracker.cpp:200-321,380-457): a sparse linear least-squares problem for the positions of
// the motion-mesh vertices -- temporal rows (pull towards the previous solution), similarity-preserving triangle
// rows on a subset of unit quads and 3x3 quads, and two barycentric rows per tracked feature.
//
// The reference solves it with Eigen::LeastSquaresConjugateGradient (warm started, tolerance eps_float, <= 2n
// iterations; Eigen is not in /root/reference).  SURVEY.md Appendix A.9 fixes OUR specification: the exact
// least-squares minimiser via the normal equations N x = g, N = A^T A:
//   S1  static rows (temporal + smoothness) exactly as generate_mesh_constraints emits them (float coefficients);
//   S2  Ns = sum over static rows of a_i * a_j in binary64, rows in emission order (constant per configuration);
//   S3  feature rows contribute w_a * w_b (exact binary64 product of two floats) quantised to Q32 fixed point
//       (llrint(p * 2^32)) and summed as exact integers -- order independent; same for g += w * dst;
//   S4  N = Ns + Q32 sums * 2^-32, plus a 1e-6 ridge on the diagonal; g = ts_rows * (ts_now * previous) + Q32 sums;
//   S5  banded root-free factorisation N = L D L^T (half bandwidth 2*(3*cols+3)+1), right-looking, pivots as reciprocals (one
//       division per column, everything else multiplications: the dependent chain per column is one division, one product and one
//       multiply-subtract -- half of Cholesky's sqrt + division, which is what bounds a GPU implementation), forward substitution
//       carried along, then D and the column-oriented backward substitution; each entry updated in pivot order, every update ONE
//       fused multiply-subtract fma(-l, c, x) (half the dependent instructions of a product and a difference; binary64);
//   S5' (round 3) NESTED DISSECTION for meshes of 8..16 columns and >= 9 rows (the OBS vector-field preset's 16 x 16): the only constraints
//       that couple vertex rows more than one row apart are the 3 x 3 quads, which start at rows 0, 4, 8, ... and end three rows further
//       down (FrameTracker.cpp:410-426) -- no constraint couples a row above row 4k with a row below it.  So the rows 4, 8, 12, ... are
//       SEPARATORS (32 unknowns each for 16 columns); the blocks between them -- rows 0..3, 5..7, 9..11, 13..15 -- are independent once the
//       separators are held back: each block is eliminated on its own (S5's right-looking band L D L^T in the order [block rows, separator
//       above, separator below], stopped after the block's own pivots; every coupling stays within the same 3-row band), the trailing
//       separator x separator windows of the blocks are ADDED (ascending block order; the original separator entries and right-hand side go
//       into the block ABOVE the separator, the others start from +0) into the separator system (block tridiagonal, half bandwidth
//       4 cols - 1), which is factorised and solved with S5 itself; then every block substitutes backwards with its separators' values
//       given.  Same minimiser (tests/test_mesh_lstsq.py: 1e-6 of numpy's lstsq); the dependent pivot chain that bounds the GPU
//       implementation shrinks from 512 to 128 + 96 -- four workgroups instead of one for most of it.
//   S6  the solution is stored as float (Eigen::VectorXf m_OptimizedMesh); inlier test and offsets as the reference.
#include "lvk_oracle.h"

#include <cmath>
#include <cstring>
#include <vector>
#include <algorithm>

namespace {

struct Triplet { int row, col; float val; };

// S5: root-free banded factorisation M = L D L^T (L unit lower triangular) of the n x n band matrix M (row layout: entry (i, j),
// i - hb <= j <= i, at M[i * (hb + 1) + (i - j)]), right-looking, pivots taken as reciprocals, with the forward substitution carried along,
// stopped after the first n_elim pivots.  Column j:  r_j = 1 / M(j, j);  L(i, j) = M(i, j) * r_j;  M(i, k) = fma(-L(i, j), M(k, j), M(i, k))
// for j < k <= i in the band (M(k, j): the UNSCALED entry);  g(i) = fma(-L(i, j), g(j), g(i)).  Every entry receives its updates in pivot
// order.  On return the columns j < n_elim hold L, rcp their reciprocal pivots, and the rows / entries >= n_elim what the eliminated pivots
// left of them (the Schur complement and the reduced right-hand side).
bool eliminate(int n, int hb, int n_elim, std::vector<double>& M, std::vector<double>& g, std::vector<double>& rcp)
{
    auto B = [&](int i, int j) -> double& { return M[(size_t)i * (hb + 1) + (i - j)]; };
    std::vector<double> colraw((size_t)hb + 1);
    rcp.assign((size_t)n, 0.0);
    for (int j = 0; j < n_elim; j++)
    {
        const double d = B(j, j);
        if (!(d > 0.0)) return false;
        const double r = 1.0 / d;
        rcp[j] = r;
        const int last = std::min(n - 1, j + hb);
        for (int i = j + 1; i <= last; i++) { colraw[i - j] = B(i, j); B(i, j) = B(i, j) * r; }      // keep the unscaled column for the updates
        for (int i = j + 1; i <= last; i++)
        {
            const double lij = B(i, j);
            g[i] = std::fma(-lij, g[j], g[i]);
            for (int k = j + 1; k <= i; k++) B(i, k) = std::fma(-lij, colraw[k - j], B(i, k));
        }
    }
    return true;
}

// L^T x = w column by column over the rows n - 1 .. 0: x(j) = w(j), w(k) = fma(-L(j, k), x(j), w(k)) for the rows k of the band above j
// that were eliminated (k < n_elim; rows >= n_elim carry given values)
void back_substitute(int n, int hb, int n_elim, const std::vector<double>& M, std::vector<double>& w)
{
    for (int j = n - 1; j >= 0; j--)
    {
        const int first = std::max(0, j - hb), lastk = std::min(j, n_elim);
        for (int k = first; k < lastk; k++) w[k] = std::fma(-M[(size_t)j * (hb + 1) + (j - k)], w[j], w[k]);
    }
}

bool solve_band(int n, int hb, std::vector<double>& N, std::vector<double>& g)
{
    std::vector<double> rcp;
    if (!eliminate(n, hb, n, N, g, rcp)) return false;
    for (int j = 0; j < n; j++) g[j] = g[j] * rcp[j];          // D w = z
    back_substitute(n, hb, n, N, g);
    return true;
}

// ---- S5': nested dissection (see the header)
bool nd_applies(int cols, int rows) { return cols >= 8 && cols <= 16 && rows >= 9; }

struct NdBlock { std::vector<int> rows; int own_rows; };          // vertex rows in elimination order: the block's own rows, then its separators

std::vector<NdBlock> nd_blocks(int rows, std::vector<int>& seps)
{
    seps.clear();
    for (int r = 4; r <= rows - 1; r += 4) seps.push_back(r);
    std::vector<NdBlock> out;
    for (size_t k = 0; k <= seps.size(); k++)
    {
        const int first = k == 0 ? 0 : seps[k - 1] + 1, last = k < seps.size() ? seps[k] - 1 : rows - 1;
        if (first > last) continue;                                 // (the last row is a separator: nothing below it)
        NdBlock b;
        for (int r = first; r <= last; r++) b.rows.push_back(r);
        b.own_rows = (int)b.rows.size();
        if (k > 0) b.rows.push_back(seps[k - 1]);                   // separator above
        if (k < seps.size()) b.rows.push_back(seps[k]);             // separator below
        out.push_back(b);
    }
    return out;
}

// N: the assembled system in natural order (row layout, half bandwidth hb), g: right-hand side in, solution out
bool solve_nd(int cols, int rows, int hb, const std::vector<double>& N, std::vector<double>& g)
{
    const int W = 2 * cols;                                         // unknowns per vertex row
    std::vector<int> seps;
    const std::vector<NdBlock> blocks = nd_blocks(rows, seps);
    const int K = (int)seps.size(), ns = K * W, hbs = std::min(ns - 1, 2 * W - 1);
    auto nat = [&](int i, int j) -> double {                        // entry (i, j) of the natural matrix, 0 outside its band
        if (i < j) std::swap(i, j);
        return i - j <= hb ? N[(size_t)i * (hb + 1) + (i - j)] : 0.0;
    };
    auto sep_index = [&](int row) { for (int k = 0; k < K; k++) if (seps[k] == row) return k; return -1; };
    std::vector<double> S((size_t)ns * (hbs + 1), 0.0), gs((size_t)ns, 0.0);
    struct Done { int n, hbb, n_elim; std::vector<double> M, w; };
    std::vector<Done> done(blocks.size());
    for (size_t b = 0; b < blocks.size(); b++)
    {
        const NdBlock& blk = blocks[b];
        const int nb = (int)blk.rows.size() * W, n_elim = blk.own_rows * W, hbb = std::min(nb - 1, hb);
        Done& d = done[b];
        d.n = nb; d.hbb = hbb; d.n_elim = n_elim;
        d.M.assign((size_t)nb * (hbb + 1), 0.0); d.w.assign((size_t)nb, 0.0);
        auto natural_of = [&](int pos) { return blk.rows[pos / W] * W + pos % W; };
        // a separator's own entries and right-hand side belong to the block ABOVE it (its "separator below" slot)
        auto owned = [&](int pos) { const int slot = pos / W; return slot < blk.own_rows || (blk.rows[slot] > blk.rows[0]); };
        for (int i = 0; i < nb; i++)
        {
            for (int j = std::max(0, i - hbb); j <= i; j++)
            {
                const bool sep_sep = i >= n_elim && j >= n_elim;
                d.M[(size_t)i * (hbb + 1) + (i - j)] = (sep_sep && !(owned(i) && owned(j))) ? 0.0 : nat(natural_of(i), natural_of(j));
            }
            d.w[i] = (i >= n_elim && !owned(i)) ? 0.0 : g[natural_of(i)];
            // (the order keeps every coupling inside the band: nothing of the system may fall outside it)
            for (int j = 0; j < i - hbb; j++) if (nat(natural_of(i), natural_of(j)) != 0.0) return false;
        }
        std::vector<double> rcp;
        if (!eliminate(nb, hbb, n_elim, d.M, d.w, rcp)) return false;
        for (int j = 0; j < n_elim; j++) d.w[j] = d.w[j] * rcp[j];
        // the trailing window into the separator system, ascending block order
        for (int i = n_elim; i < nb; i++)
        {
            const int si = sep_index(blk.rows[i / W]) * W + i % W;
            gs[si] = gs[si] + d.w[i];
            for (int j = n_elim; j <= i; j++)
            {
                const int sj = sep_index(blk.rows[j / W]) * W + j % W;
                const int hi = std::max(si, sj), lo = std::min(si, sj);
                S[(size_t)hi * (hbs + 1) + (hi - lo)] = S[(size_t)hi * (hbs + 1) + (hi - lo)] + d.M[(size_t)i * (hbb + 1) + (i - j)];
            }
        }
    }
    if (!solve_band(ns, hbs, S, gs)) return false;
    for (size_t b = 0; b < blocks.size(); b++)
    {
        const NdBlock& blk = blocks[b];
        Done& d = done[b];
        for (int i = d.n_elim; i < d.n; i++) d.w[i] = gs[sep_index(blk.rows[i / W]) * W + i % W];
        back_substitute(d.n, d.hbb, d.n_elim, d.M, d.w);
        for (int i = 0; i < d.n_elim; i++) g[blk.rows[i / W] * W + i % W] = d.w[i];
    }
    for (int k = 0; k < K; k++) for (int c = 0; c < W; c++) g[seps[k] * W + c] = gs[k * W + c];
    return true;
}

} // namespace


struct lvko_mesh_solver
{
    int cols = 0, rows = 0, n = 0, hb = 0;
    float ts_gen = 0;                                  // temporal weight the static rows were generated with
    std::vector<double> Ns;                            // lower band, n x (hb+1): Ns[i*(hb+1) + (i-j)]
    std::vector<float> mesh;                           // m_OptimizedMesh (2 floats per vertex, absolute tracking-frame coordinates)
    int static_rows = 0;
    std::vector<Triplet> static_triplets;

    double& band(std::vector<double>& B, int i, int j) { return B[(size_t)i * (hb + 1) + (i - j)]; }
};

extern "C" {

// generate_mesh_constraints (FrameTracker.cpp:380-457) for a mesh of cols x rows vertices over a region of
// gen_region_w x gen_region_h tracking pixels (only its key aspect ratio matters), with the smoothing weights in
// force when the reference (re)generates the constraints (FrameTracker.cpp:74-82 -- note: the PREVIOUS settings').
lvko_mesh_solver* lvko_mesh_solver_create(int cols, int rows, float gen_region_w, float gen_region_h, float temporal_smoothing, float local_smoothing)
{
    auto* s = new lvko_mesh_solver();
    s->cols = cols; s->rows = rows; s->n = 2 * cols * rows;
    s->hb = std::min(s->n - 1, 2 * (3 * cols + 3) + 1);
    // VirtualGrid(mesh_size, Rect2f(tl, (Size2f(mesh_size) / Size2f(grid_size)) * region.size())): key size in float
    const float gaw = ((float)cols / (float)(cols - 1)) * gen_region_w, gah = ((float)rows / (float)(rows - 1)) * gen_region_h;
    const float gen_key_w = gaw / (float)cols, gen_key_h = gah / (float)rows;
    s->ts_gen = temporal_smoothing;
    s->mesh.assign((size_t)s->n, 0.0f);                // Eigen::VectorXf::Zero

    std::vector<Triplet>& T = s->static_triplets;
    int row = 0;
    for (int index = 0; index < cols * rows; index++)
    {
        T.push_back({row++, 2 * index, temporal_smoothing});
        T.push_back({row++, 2 * index + 1, temporal_smoothing});
    }
    // v1 = -key_size.aspectRatio() (double), v2 = -1.0 / v1
    const double v1 = -((double)gen_key_w / (double)gen_key_h), v2 = -1.0 / v1;
    for (int r = 0, index = 0; r < rows; r++)
        for (int c = 0; c < cols; c++, index++)
        {
            int quad = 1;
            if (c % 4 == 0 && r % 4 == 0) quad = 3;
            else if ((c + r) % 2 != 1 && c != 0 && r != 0 && c != cols - 2 && r != rows - 2) continue;
            if (c >= cols - quad || r >= rows - quad) continue;
            const int i00 = 2 * index, i10 = i00 + 2 * quad;
            const int i01 = 2 * (index + quad * cols), i11 = i01 + 2 * quad;
            const float weight = local_smoothing;
            const float w1 = (float)(v1 * weight), w2 = (float)(v2 * weight);
            T.push_back({row, i00, -weight}); T.push_back({row, i01, weight}); T.push_back({row, i01 + 1, -w2}); T.push_back({row, i11 + 1, w2}); row++;
            T.push_back({row, i00 + 1, -weight}); T.push_back({row, i01, w2}); T.push_back({row, i01 + 1, weight}); T.push_back({row, i11, -w2}); row++;
            T.push_back({row, i00, -weight}); T.push_back({row, i10, weight}); T.push_back({row, i10 + 1, -w1}); T.push_back({row, i11 + 1, w1}); row++;
            T.push_back({row, i00 + 1, -weight}); T.push_back({row, i10, w1}); T.push_back({row, i10 + 1, weight}); T.push_back({row, i11, -w1}); row++;
        }
    s->static_rows = row;

    // S2: Ns = sum of row outer products (rows in emission order)
    s->Ns.assign((size_t)s->n * (s->hb + 1), 0.0);
    for (size_t a = 0; a < T.size();)
    {
        size_t b = a;
        while (b < T.size() && T[b].row == T[a].row) b++;
        for (size_t p = a; p < b; p++)
            for (size_t q = a; q < b; q++)
                if (T[p].col >= T[q].col)
                    s->band(s->Ns, T[p].col, T[q].col) = s->band(s->Ns, T[p].col, T[q].col) + (double)T[p].val * (double)T[q].val;
        a = b;
    }
    return s;
}

void lvko_mesh_solver_destroy(lvko_mesh_solver* s) { delete s; }
void lvko_mesh_solver_reset(lvko_mesh_solver* s) { std::fill(s->mesh.begin(), s->mesh.end(), 0.0f); }     // FrameTracker::restart :103
int lvko_mesh_solver_static_rows(const lvko_mesh_solver* s) { return s->static_rows; }
int lvko_mesh_solver_static_triplets(const lvko_mesh_solver* s) { return (int)s->static_triplets.size(); }
const float* lvko_mesh_solver_mesh(const lvko_mesh_solver* s) { return s->mesh.data(); }

// estimate_local_motions (FrameTracker.cpp:200-321).  tracked/matched: n_pts x 2 floats.  Writes inlier flags and the
// normalised offsets (cols*rows*2 floats) of the motion mesh.  temporal_now = the CURRENT settings' temporal_smoothing
// (used for the right-hand side, :229-230), threshold = acceptance_threshold, (region_w, region_h) = tracking resolution.
int lvko_mesh_solver_solve(lvko_mesh_solver* s, const float* tracked, const float* matched, int n_pts,
                           float region_w, float region_h, float temporal_now, float threshold, uint8_t* inliers, float* offsets)
{
    const int n = s->n, hb = s->hb, W = s->cols;
    // mesh_grid of estimate_local_motions (:211-216), from the CURRENT tracking region
    const float aw = ((float)s->cols / (float)(s->cols - 1)) * region_w, ah = ((float)s->rows / (float)(s->rows - 1)) * region_h;
    const float key_w = aw / (float)s->cols, key_h = ah / (float)s->rows;
    std::vector<double> N = s->Ns, g((size_t)n, 0.0);
    std::vector<long long> Nq((size_t)n * (hb + 1), 0), gq((size_t)n, 0);
    std::vector<int> fi((size_t)n_pts * 4);
    std::vector<float> fw((size_t)n_pts * 4);
    const double Q = 4294967296.0;

    // temporal right-hand side
    for (int i = 0; i < n; i++) g[i] = (double)s->ts_gen * (double)(temporal_now * s->mesh[i]);

    for (int f = 0; f < n_pts; f++)
    {
        const float sx = tracked[2 * f], sy = tracked[2 * f + 1];
        int kx = (int)(size_t)((sx - 0.0f) / key_w), ky = (int)(size_t)((sy - 0.0f) / key_h);        // VirtualGrid::key_of
        kx = std::min(std::max(kx, 0), s->cols - 1); ky = std::min(std::max(ky, 0), s->rows - 1);          // clamp to grid_size (:243-244)
        const int i00 = 2 * (ky * W + kx), i11 = 2 * ((ky + 1) * W + (kx + 1));
        const int i10 = i00 + 2, i01 = i11 - 2;
        // barycentric_rect({key_to_point(k00), key_to_point(k11)}, src) (Functions/Math.tpp:247-265), float
        const float x1 = (float)kx * key_w, y1 = (float)ky * key_h;
        const float x2r = (float)(kx + 1) * key_w, y2r = (float)(ky + 1) * key_h;
        const float rw = x2r - x1, rh = y2r - y1;                   // cv::Rect_(pt1, pt2): width = max - min
        const float inv = 1.0f / (rw * rh);
        const float x2 = x1 + rw, y2 = y1 + rh;
        const float rx1 = x2 - sx, ry1 = y2 - sy, rx2 = sx - x1, ry2 = sy - y1;
        const float w[4] = {(float)(double)(rx1 * ry1 * inv), (float)(double)(rx1 * ry2 * inv), (float)(double)(rx2 * ry2 * inv), (float)(double)(rx2 * ry1 * inv)};
        const int idx[4] = {i00, i01, i11, i10};
        if (i11 + 1 >= n) return -1;                                // a point in the last grid cell row/col would index past the mesh
        for (int a = 0; a < 4; a++) { fi[4 * f + a] = idx[a]; fw[4 * f + a] = w[a]; }
        for (int comp = 0; comp < 2; comp++)
        {
            const float dst = matched[2 * f + comp];
            for (int a = 0; a < 4; a++)
            {
                const int ia = idx[a] + comp;
                gq[ia] += llrint((double)w[a] * (double)dst * Q);
                for (int b = 0; b < 4; b++)
                {
                    const int ib = idx[b] + comp;
                    if (ia >= ib) Nq[(size_t)ia * (hb + 1) + (ia - ib)] += llrint((double)w[a] * (double)w[b] * Q);
                }
            }
        }
    }
    for (size_t k = 0; k < N.size(); k++) N[k] = N[k] + (double)Nq[k] / Q;
    for (int i = 0; i < n; i++) { g[i] = g[i] + (double)gq[i] / Q; N[(size_t)i * (hb + 1)] = N[(size_t)i * (hb + 1)] + 1e-6; }

    if (nd_applies(s->cols, s->rows)) { if (!solve_nd(s->cols, s->rows, hb, N, g)) return -2; }
    else if (!solve_band(n, hb, N, g)) return -2;
    for (int i = 0; i < n; i++) s->mesh[i] = (float)g[i];

    // inlier status (:279-310): L1 reprojection error of each feature through its quad
    for (int f = 0; f < n_pts; f++)
    {
        const int* id = &fi[4 * f]; const float* w = &fw[4 * f];
        const float x = w[0] * s->mesh[id[0]] + w[1] * s->mesh[id[1]] + w[2] * s->mesh[id[2]] + w[3] * s->mesh[id[3]];
        const float y = w[0] * s->mesh[id[0] + 1] + w[1] * s->mesh[id[1] + 1] + w[2] * s->mesh[id[2] + 1] + w[3] * s->mesh[id[3] + 1];
        const float err = std::fabs(x - matched[2 * f]) + std::fabs(y - matched[2 * f + 1]);
        inliers[f] = err < threshold ? 1 : 0;
    }
    // offsets (:316-320): (aligned grid - solved) / region size
    for (int r = 0, index = 0; r < s->rows; r++)
        for (int c = 0; c < s->cols; c++, index++)
        {
            offsets[2 * index] = ((float)c * key_w - s->mesh[2 * index]) / region_w;
            offsets[2 * index + 1] = ((float)r * key_h - s->mesh[2 * index + 1]) / region_h;
        }
    return 0;
}

} // extern "C"
