// The device mesh solver as an object: constraint generation (FrameTracker.cpp:380-457), the nested-dissection layout (oracle S5', DESIGN.md
// section 4), the launch sequence of mesh.hip's kernels, and the per-stage C-ABI entry points (lvk_hip_mesh_solver_*, include/lvk_hip.h).
// Reference: FrameTracker::estimate_local_motions (LiveVisionKit/Vision/FrameTracker.cpp:200-321).
#include "mesh_internal.hpp"
#include "host_logic.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace lvkmesh;

// ---- host side ------------------------------------------------------------------------------------------------------------------
struct lvk_mesh_solver_dev
{
    lvk_hip_ctx* ctx = nullptr;
    int cols = 0, rows = 0, n = 0, hb = 0;
    float ts_gen = 0.0f;
    double* d_stat = nullptr; long long* d_acc = nullptr;   // d_acc: Nq (n * ld) then gq (n), one allocation, one memset per solve
    float* d_mesh = nullptr; double* d_Lc = nullptr; double* d_N = nullptr;      // d_N: band then right-hand side
    int* d_flags = nullptr;
    bool generic = false;                   // k_mesh_solve_generic / k_mesh_backsolve_generic (meshes outside the register-window solver's shapes)
    // nested dissection (8 .. 16 columns, >= 9 rows): the blocks' band systems and the separator system, all built once per configuration
    bool nd = false;
    int nblocks = 0, ns = 0, hbs = 0, s_entries = 0;
    bool sep_generic = false;
    MeshBlockDev* d_blocks = nullptr;
    double* d_nd = nullptr;                 // one allocation: per block N | g | wz | Lc | Rc | T, then the separator system S | gs | wzs | Lcs | xs, then X
    int* d_ndi = nullptr;                   // one allocation: ndst | gdst | ssrc | gsrc | sep_nat | per block xs_of, nat_of
    size_t off_Nall = 0, off_wzall = 0, off_Tall = 0, off_S = 0, off_gs = 0, off_wzs = 0, off_Lcs = 0, off_xs = 0, off_X = 0;
    size_t ioff_ndst = 0, ioff_gdst = 0, ioff_ssrc = 0, ioff_gsrc = 0, ioff_sepnat = 0;
    unsigned* d_ticket = nullptr;
};

static bool nd_applies(int cols, int rows) { return cols >= 8 && cols <= 16 && rows >= 9; }     // oracle S5' (the register-window kernels' column range)

void lvk_mesh_solver_free(lvk_mesh_solver_dev* s)
{
    if (!s) return;
    void* dev[] = {s->d_stat, s->d_acc, s->d_mesh, s->d_Lc, s->d_N, s->d_flags, s->d_blocks, s->d_nd, s->d_ndi, s->d_ticket};
    for (void* p : dev) if (p) (void)hipFree(p);
    delete s;
}

// generate_mesh_constraints for a cols x rows mesh (FrameTracker.cpp:380-457): the static band is built on the host and uploaded once
int lvk_mesh_solver_create(lvk_hip_ctx* ctx, int cols, int rows, float gen_w, float gen_h, float temporal, float local, lvk_mesh_solver_dev** out)
{
    LVK_HIP_REQUIRE(ctx, out != nullptr && cols >= 2 && rows >= 2);
    *out = nullptr;
    lvkh::MeshSolverH host;
    host.generate(cols, rows, gen_w, gen_h, temporal, local);
    // the register-window solver: meshes up to 16 columns (half bandwidth 103) and 16 x 64 vertices, every band with a tile that hands its
    // columns to the chain; everything else (17 x 17, 32 x 32, ...) takes the generic kernels -- same specification, same bits
    const bool force_generic = std::getenv("LVK_HIP_MESH_GENERIC") != nullptr;      // tests: the generic kernels on the preset's mesh
    const bool fast = host.hb() <= MS_HB_MAX && host.n() <= MS_N_MAX && host.n() >= 4 && band_groups(host.hb(), (host.hb() + MS_TB) / MS_TB - 1) >= 1;
    LVK_HIP_REQUIRE(ctx, host.hb() <= MG_HB_MAX);                           // motion_resolution beyond 167 columns
    auto* s = new lvk_mesh_solver_dev();
    s->ctx = ctx; s->cols = cols; s->rows = rows; s->n = host.n(); s->hb = host.hb(); s->ts_gen = temporal;
    s->generic = !fast || force_generic;
    const size_t band = (size_t)s->n * (s->hb + 1);
    auto fail = [&](hipError_t e) { lvk_mesh_solver_free(s); return ctx->fail(LVK_HIP_ERR_RUNTIME, hipGetErrorString(e)); };
    hipError_t e;
    if ((e = hipMalloc((void**)&s->d_stat, band * sizeof(double))) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void**)&s->d_acc, (band + s->n) * sizeof(long long))) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void**)&s->d_mesh, s->n * sizeof(float))) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void**)&s->d_Lc, (band + MS_NT) * sizeof(double))) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void**)&s->d_N, (band + 2 * (size_t)s->n) * sizeof(double))) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void**)&s->d_flags, sizeof(int))) != hipSuccess) return fail(e);
    if ((e = hipMemcpy(s->d_stat, host.static_band().data(), band * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
    if ((e = hipMemset(s->d_mesh, 0, s->n * sizeof(float))) != hipSuccess) return fail(e);
    if ((e = hipMemset(s->d_Lc, 0, (band + MS_NT) * sizeof(double))) != hipSuccess) return fail(e);
    if ((e = hipMemset(s->d_acc, 0, (band + s->n) * sizeof(long long))) != hipSuccess) return fail(e);       // kept clear by k_mesh_prepare
    if ((e = hipMemset(s->d_flags, 0, sizeof(int))) != hipSuccess) return fail(e);                            // kept clear by k_mesh_solve
    if (nd_applies(cols, rows) && !force_generic)
    {
        // ---- nested dissection (oracle S5'): separator rows 4, 8, ...; blocks = the rows between them, each followed by its separators
        const int W = 2 * cols, n = s->n, hb = s->hb, ld = hb + 1;
        std::vector<int> seps;
        for (int r = 4; r <= rows - 1; r += 4) seps.push_back(r);
        const int K = (int)seps.size();
        struct Blk { std::vector<int> rows; int own; int n, hb, n_elim, ns; size_t N, g, wz, Lc, Rc, T; size_t ixs, inat; };
        std::vector<Blk> blks;
        for (int k = 0; k <= K; k++)
        {
            const int first = k == 0 ? 0 : seps[k - 1] + 1, last = k < K ? seps[k] - 1 : rows - 1;
            if (first > last) continue;
            Blk b{};
            for (int r = first; r <= last; r++) b.rows.push_back(r);
            b.own = (int)b.rows.size();
            if (k > 0) b.rows.push_back(seps[k - 1]);
            if (k < K) b.rows.push_back(seps[k]);
            b.n = (int)b.rows.size() * W; b.n_elim = b.own * W; b.ns = b.n - b.n_elim; b.hb = std::min(b.n - 1, hb);
            blks.push_back(b);
        }
        s->nblocks = (int)blks.size(); s->ns = K * W; s->hbs = std::min(s->ns - 1, 2 * W - 1);
        bool blocks_fit = true;
        for (const Blk& b : blks)
            blocks_fit = blocks_fit && b.ns <= 255 && b.hb <= MS_HB_MAX && b.n <= MS_N_MAX && b.n >= 4 && band_groups(b.hb, (b.hb + MS_TB) / MS_TB - 1) >= 1;
        if (!blocks_fit) { lvk_mesh_solver_free(s); return ctx->fail(LVK_HIP_ERR_ARG, "mesh solver: a block of the nested dissection does not fit the register-window kernels"); }
        s->sep_generic = !(s->hbs <= MS_HB_MAX && s->ns <= MS_N_MAX && s->ns >= 4 && band_groups(s->hbs, (s->hbs + MS_TB) / MS_TB - 1) >= 1);
        if (s->sep_generic && s->hbs > MG_HB_MAX) { lvk_mesh_solver_free(s); return ctx->fail(LVK_HIP_ERR_ARG, "mesh solver: separator system too wide"); }
        // layout of the binary64 arena
        size_t at = 0;
        auto take = [&](size_t count) { const size_t o = at; at += (count + 1) & ~(size_t)1; return o; };
        s->off_Nall = at;
        for (Blk& b : blks) b.N = take((size_t)b.n * (b.hb + 1));
        for (Blk& b : blks) b.g = take(b.n);
        s->off_wzall = at;
        for (Blk& b : blks) b.wz = take(b.n);
        for (Blk& b : blks) b.Lc = take((size_t)b.n * (b.hb + 1) + MS_NT);
        for (Blk& b : blks) b.Rc = take((size_t)b.n * (b.hb + 1) + MS_NT);
        s->off_Tall = at;
        for (Blk& b : blks) b.T = take((size_t)b.ns * b.ns);
        const size_t lds = s->hbs + 1;
        s->off_S = take((size_t)s->ns * lds); s->off_gs = take(s->ns); s->off_wzs = take(s->ns); s->off_Lcs = take((size_t)s->ns * lds + MS_NT);
        s->off_xs = take(s->ns); s->off_X = take(n);
        const size_t doubles = at;
        // index tables
        std::vector<int> tab;
        auto itake = [&](size_t count) { const size_t o = tab.size(); tab.resize(o + count, -1); return o; };
        s->ioff_ndst = itake((size_t)n * ld); s->ioff_gdst = itake(n);
        s->s_entries = s->ns * (int)lds;
        s->ioff_ssrc = itake(2 * (size_t)s->s_entries); s->ioff_gsrc = itake(2 * (size_t)s->ns); s->ioff_sepnat = itake(s->ns);
        for (Blk& b : blks) { b.ixs = itake(std::max(b.ns, 1)); b.inat = itake(b.n_elim); }
        // where every vertex row sits: (block, slot) of its owning block (separators: the block ABOVE), and the slots it has as a separator
        auto sep_index = [&](int row) { for (int k = 0; k < K; k++) if (seps[k] == row) return k; return -1; };
        auto slot_of = [&](const Blk& b, int row) { for (size_t q = 0; q < b.rows.size(); q++) if (b.rows[q] == row) return (int)q; return -1; };
        auto owner_of = [&](int row) {                                    // block whose band holds the row's own entries and right-hand side
            for (int bi = 0; bi < (int)blks.size(); bi++)
            {
                const int q = slot_of(blks[bi], row);
                if (q < 0) continue;
                if (q < blks[bi].own || row > blks[bi].rows[0]) return bi;    // an own row, or the separator BELOW the block
            }
            return -1;
        };
        const std::vector<double>& stat = host.static_band();
        for (int k = 0; k < n; k++)
            for (int t = 0; t <= hb && k + t < n; t++)
            {
                const int i = k + t, ri = i / W, rk = k / W;
                // the block that holds both: an own row decides; two rows of the same separator go to its owner
                int bi = -1;
                for (int cand = 0; cand < (int)blks.size() && bi < 0; cand++)
                {
                    const int qi = slot_of(blks[cand], ri), qk = slot_of(blks[cand], rk);
                    if (qi < 0 || qk < 0) continue;
                    const bool own_i = qi < blks[cand].own, own_k = qk < blks[cand].own;
                    if (own_i || own_k) bi = cand;
                    else if (ri == rk && owner_of(ri) == cand) bi = cand;
                }
                const size_t src = (size_t)k * ld + t;
                if (bi < 0)
                {
                    if (stat[src] != 0.0) { lvk_mesh_solver_free(s); return ctx->fail(LVK_HIP_ERR_RUNTIME, "mesh solver: a constraint couples two blocks of the nested dissection"); }
                    continue;
                }
                const Blk& b = blks[bi];
                const int pi = slot_of(b, ri) * W + i % W, pk = slot_of(b, rk) * W + k % W;
                const int hi = std::max(pi, pk), lo = std::min(pi, pk);
                if (hi - lo > b.hb)
                {
                    if (stat[src] != 0.0) { lvk_mesh_solver_free(s); return ctx->fail(LVK_HIP_ERR_RUNTIME, "mesh solver: a constraint leaves a block's band"); }
                    continue;
                }
                tab[s->ioff_ndst + src] = (int)(b.N - s->off_Nall + (size_t)lo * (b.hb + 1) + (hi - lo));
            }
        for (int i = 0; i < n; i++)
        {
            const int bi = owner_of(i / W);
            tab[s->ioff_gdst + i] = (int)(blks[bi].g - s->off_Nall + (size_t)slot_of(blks[bi], i / W) * W + i % W);      // (g follows N in the arena)
        }
        for (int bi = 0; bi < (int)blks.size(); bi++)
        {
            const Blk& b = blks[bi];
            for (int p = 0; p < b.n_elim; p++) tab[b.inat + p] = b.rows[p / W] * W + p % W;
            for (int q = 0; q < b.ns; q++) tab[b.ixs + q] = sep_index(b.rows[(b.n_elim + q) / W]) * W + q % W;
            // this block's trailing window into the separator system (blocks in ascending order fill source 0, then source 1)
            for (int li = 0; li < b.ns; li++)
            {
                const int si = tab[b.ixs + li];
                int* gs = &tab[s->ioff_gsrc + 2 * (size_t)si];
                gs[gs[0] < 0 ? 0 : 1] = (int)(b.wz - s->off_wzall + b.n_elim + li);
                for (int lk = 0; lk <= li; lk++)
                {
                    const int sk = tab[b.ixs + lk];                         // (the separator above comes first: sk <= si)
                    int* ss = &tab[s->ioff_ssrc + 2 * ((size_t)sk * lds + (si - sk))];
                    ss[ss[0] < 0 ? 0 : 1] = (bi << 16) | (li << 8) | lk;
                }
            }
        }
        for (int k = 0; k < K; k++) for (int c = 0; c < W; c++) tab[s->ioff_sepnat + (size_t)k * W + c] = seps[k] * W + c;
        if ((e = hipMalloc((void**)&s->d_nd, doubles * sizeof(double))) != hipSuccess) return fail(e);
        if ((e = hipMemset(s->d_nd, 0, doubles * sizeof(double))) != hipSuccess) return fail(e);       // structural zeros of the blocks' bands, columns of L beyond n_elim
        if ((e = hipMalloc((void**)&s->d_ndi, tab.size() * sizeof(int))) != hipSuccess) return fail(e);
        if ((e = hipMemcpy(s->d_ndi, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
        std::vector<MeshBlockDev> hb_(blks.size());
        for (size_t bi = 0; bi < blks.size(); bi++)
        {
            const Blk& b = blks[bi];
            hb_[bi] = MeshBlockDev{b.n, b.hb, b.n_elim, (b.hb + MS_TB) / MS_TB, s->d_nd + b.N, s->d_nd + b.g, s->d_nd + b.wz, s->d_nd + b.Lc, s->d_nd + b.Rc,
                                   b.ns > 0 ? s->d_nd + b.T : nullptr, s->d_ndi + b.ixs, s->d_ndi + b.inat};
        }
        if ((e = hipMalloc((void**)&s->d_blocks, hb_.size() * sizeof(MeshBlockDev))) != hipSuccess) return fail(e);
        if ((e = hipMemcpy(s->d_blocks, hb_.data(), hb_.size() * sizeof(MeshBlockDev), hipMemcpyHostToDevice)) != hipSuccess) return fail(e);
        if ((e = hipMalloc((void**)&s->d_ticket, sizeof(unsigned))) != hipSuccess) return fail(e);
        if ((e = hipMemset(s->d_ticket, 0, sizeof(unsigned))) != hipSuccess) return fail(e);
        s->nd = true; s->generic = false;
    }
    *out = s;
    return LVK_HIP_OK;
}

int lvk_mesh_solver_reset(lvk_mesh_solver_dev* s, hipStream_t stream)      // FrameTracker::restart: m_OptimizedMesh = 0 (:103)
{
    LVK_HIP_CHECK(s->ctx, hipMemsetAsync(s->d_mesh, 0, s->n * sizeof(float), stream));
    return LVK_HIP_OK;
}

int lvk_mesh_solver_cols(const lvk_mesh_solver_dev* s) { return s->cols; }
int lvk_mesh_solver_rows(const lvk_mesh_solver_dev* s) { return s->rows; }

// d_scratch: 8 x 4 bytes per pair (the pair's unknown indices and weights, kept between the kernels).  d_p1 / d_p2: tracked / matched points; d_count: pair count decided on the GPU (or nullptr: n_pts pairs).  Results are written to
// device-visible host memory: offsets (cols * rows * 2 floats), inlier flags, status (0 ok, 1 fewer than min_samples pairs, 2 a feature
// outside the mesh, 3 factorisation broke down); for a status != 0 the previous solution is left untouched.
int lvk_launch_mesh_solve(lvk_mesh_solver_dev* s, hipStream_t stream, void* d_scratch, const float2* d_p1, const float2* d_p2, const int* d_count, int n_pts,
                          int min_samples, float region_w, float region_h, float temporal_now, float threshold,
                          float* h_offsets, uint8_t* h_mask, int* h_status)
{
    lvk_hip_ctx* ctx = s->ctx;
    LVK_HIP_REQUIRE(ctx, n_pts >= 0 && d_scratch != nullptr);
    const size_t band = (size_t)s->n * (s->hb + 1);
    MeshArgs a;
    a.cols = s->cols; a.rows = s->rows; a.n = s->n; a.hb = s->hb; a.nbands = (s->hb + MS_TB) / MS_TB;
    a.stat = s->d_stat; a.Nq = s->d_acc; a.gq = s->d_acc + band; a.N = s->d_N; a.g0 = s->d_N + band; a.wz = s->d_N + band + s->n; a.mesh = s->d_mesh; a.Lc = s->d_Lc;
    a.fidx = (int*)d_scratch; a.fw = (float*)d_scratch + 4 * (size_t)std::max(n_pts, 1); a.p1 = d_p1; a.p2 = d_p2; a.count = d_count; a.n_pts = n_pts; a.min_samples = min_samples;
    a.region_w = region_w; a.region_h = region_h; a.ts_gen = s->ts_gen; a.ts_now = temporal_now; a.threshold = threshold;
    a.flags = s->d_flags; a.out_offsets = h_offsets; a.out_mask = h_mask; a.out_status = h_status;
    a.nd = 0; a.nblocks = 0; a.n_elim = s->n; a.n_nat = s->n; a.blocks = nullptr; a.Rc = nullptr; a.T = nullptr; a.ndst = nullptr; a.gdst = nullptr;
    a.sep_wz = nullptr; a.sep_Lc = nullptr; a.sep_hb = 0; a.fuse_sep = 0;
    a.ssrc = nullptr; a.gsrc = nullptr; a.s_entries = 0; a.ns = 0; a.Tall = nullptr; a.wzall = nullptr; a.xs = nullptr; a.X = nullptr; a.sep_nat = nullptr; a.ticket = nullptr;
    if (n_pts > 0) launch_assemble((unsigned)((n_pts + 127) / 128), stream, a);
    if (s->nd)
    {
        // nested dissection: scatter into the blocks' bands | the blocks side by side | separator system | the blocks' backward substitutions
        // side by side, the last one to finish runs phase 3
        MeshArgs p = a;
        p.nd = 1; p.ndst = s->d_ndi + s->ioff_ndst; p.gdst = s->d_ndi + s->ioff_gdst; p.N = s->d_nd + s->off_Nall; p.g0 = s->d_nd + s->off_Nall;
        launch_prepare(stream, p);
        MeshArgs f = a;
        f.nd = 1; f.blocks = s->d_blocks; f.nblocks = s->nblocks; f.xs = s->d_nd + s->off_xs; f.X = s->d_nd + s->off_X; f.sep_nat = s->d_ndi + s->ioff_sepnat;
        f.ns = s->ns; f.ticket = s->d_ticket;
        launch_solve((unsigned)s->nblocks, stream, f);
        MeshArgs q = a;
        q.nd = 1; q.n = s->ns; q.hb = s->hbs; q.nbands = (s->hbs + MS_TB) / MS_TB; q.n_elim = s->ns;
        q.N = s->d_nd + s->off_S; q.g0 = s->d_nd + s->off_gs; q.wz = s->d_nd + s->off_wzs; q.Lc = s->d_nd + s->off_Lcs; q.xs = s->d_nd + s->off_xs;
        q.ssrc = s->d_ndi + s->ioff_ssrc; q.gsrc = s->d_ndi + s->ioff_gsrc; q.s_entries = s->s_entries; q.ns = s->ns;
        q.Tall = s->d_nd + s->off_Tall; q.wzall = s->d_nd + s->off_wzall;
        q.blocks = s->d_blocks;
        if ((2 * s->cols) % NT_TILE == 0)
        {
            const int tiles_per_col = s->hbs / NT_TILE + 1;
            launch_sep_assemble_tiled((unsigned)((s->ns / NT_TILE) * tiles_per_col + 1), stream, q, tiles_per_col);
        }
        else launch_sep_assemble((unsigned)((s->s_entries + s->ns + 63) / 64), stream, q);
        q.blocks = nullptr;
        if (s->sep_generic)
        {
            launch_solve_generic(stream, q);
            launch_backsolve_generic(stream, q);
        }
        else
        {
            launch_solve(1u, stream, q);
            f.fuse_sep = 1; f.sep_wz = q.wz; f.sep_Lc = q.Lc; f.sep_hb = s->hbs;      // its backward substitution: inside the blocks' kernel
        }
        launch_backsolve((unsigned)s->nblocks, stream, f);
        LVK_HIP_CHECK(ctx, hipGetLastError());
        return LVK_HIP_OK;
    }
    launch_prepare(stream, a);
    if (s->generic)
    {
        launch_solve_generic(stream, a);
        launch_backsolve_generic(stream, a);
    }
    else
    {
        launch_solve(1u, stream, a);
        launch_backsolve(1u, stream, a);
    }
    LVK_HIP_CHECK(ctx, hipGetLastError());
    return LVK_HIP_OK;
}

// ---- C-ABI: the solver on its own (per-stage entry point, lvk_hip.h) ----------------------------------------------------------------
struct lvk_hip_mesh_solver
{
    lvk_hip_ctx* ctx = nullptr;
    lvk_mesh_solver_dev* dev = nullptr;
    float2* d_pts = nullptr;                 // tracked | matched
    void* d_scratch = nullptr; int cap = 0;
    float* h_offsets = nullptr; uint8_t* h_mask = nullptr; int* h_status = nullptr;      // pinned
};

extern "C" {

void lvk_hip_mesh_solver_destroy(lvk_hip_mesh_solver* s)
{
    if (!s) return;
    lvk_device_guard device_guard(s->ctx);
    lvk_mesh_solver_free(s->dev);
    if (s->d_pts) (void)hipFree(s->d_pts);
    if (s->d_scratch) (void)hipFree(s->d_scratch);
    if (s->h_offsets) (void)hipHostFree(s->h_offsets);
    if (s->h_mask) (void)hipHostFree(s->h_mask);
    if (s->h_status) (void)hipHostFree(s->h_status);
    delete s;
}

int lvk_hip_mesh_solver_create(lvk_hip_ctx* ctx, int cols, int rows, float gen_region_w, float gen_region_h,
                               float temporal_smoothing, float local_smoothing, int max_points, lvk_hip_mesh_solver** out)
{
    LVK_HIP_ENTRY(ctx);
    LVK_HIP_REQUIRE(ctx, out != nullptr && max_points > 0);
    *out = nullptr;
    auto* s = new lvk_hip_mesh_solver();
    s->ctx = ctx; s->cap = max_points;
    int rc = lvk_mesh_solver_create(ctx, cols, rows, gen_region_w, gen_region_h, temporal_smoothing, local_smoothing, &s->dev);
    if (rc != LVK_HIP_OK) { delete s; return rc; }
    if (hipMalloc((void**)&s->d_pts, 2 * (size_t)max_points * sizeof(float2)) != hipSuccess ||
        hipMalloc(&s->d_scratch, 32 * (size_t)max_points) != hipSuccess ||
        hipHostMalloc((void**)&s->h_offsets, (size_t)cols * rows * 2 * sizeof(float), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&s->h_mask, (size_t)max_points, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&s->h_status, sizeof(int), hipHostMallocDefault) != hipSuccess)
    { lvk_hip_mesh_solver_destroy(s); return ctx->fail(LVK_HIP_ERR_RUNTIME, "mesh solver: allocation failed"); }
    *out = s;
    return LVK_HIP_OK;
}

int lvk_hip_mesh_solver_reset(lvk_hip_mesh_solver* s)
{
    if (!s) return LVK_HIP_ERR_ARG;
    lvk_device_guard device_guard(s->ctx);
    return lvk_mesh_solver_reset(s->dev, s->ctx->stream);
}

// Returns 0 when an estimate was produced, 2 when a point fell into the last cell row / column of the mesh, 3 when the factorisation
// broke down (both: "no estimate", the previous solution is kept -- FrameTracker.cpp:243-247), negative on errors.
int lvk_hip_mesh_solver_solve(lvk_hip_mesh_solver* s, const float* tracked, const float* matched, int n, float region_w, float region_h,
                              float temporal_now, float threshold, uint8_t* inliers, float* offsets)
{
    if (!s) return LVK_HIP_ERR_ARG;
    lvk_hip_ctx* ctx = s->ctx;
    lvk_device_guard device_guard(ctx);
    const int cap = s->cap;
    LVK_HIP_REQUIRE(ctx, tracked && matched && inliers && offsets && n >= 0 && n <= cap);
    hipStream_t st = ctx->stream;
    if (n > 0)
    {
        LVK_HIP_CHECK(ctx, hipMemcpyAsync(s->d_pts, tracked, (size_t)n * sizeof(float2), hipMemcpyHostToDevice, st));
        LVK_HIP_CHECK(ctx, hipMemcpyAsync(s->d_pts + cap, matched, (size_t)n * sizeof(float2), hipMemcpyHostToDevice, st));
    }
    const int rc = lvk_launch_mesh_solve(s->dev, st, s->d_scratch, s->d_pts, s->d_pts + cap, nullptr, n, 0, region_w, region_h, temporal_now, threshold,
                                         s->h_offsets, s->h_mask, s->h_status);
    if (rc != LVK_HIP_OK) return rc;
    LVK_HIP_CHECK(ctx, hipStreamSynchronize(st));
    if (*s->h_status != 0) return *s->h_status;
    std::memcpy(inliers, s->h_mask, (size_t)n);
    std::memcpy(offsets, s->h_offsets, (size_t)lvk_mesh_solver_cols(s->dev) * lvk_mesh_solver_rows(s->dev) * 2 * sizeof(float));
    return 0;
}

} // extern "C"
