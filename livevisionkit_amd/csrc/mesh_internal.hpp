// Shared by the two translation units of the device mesh solver (not installed):
//   mesh.hip          the kernels (assemble, prepare, band L D L^T, separator assembly, backward substitution; the generic variants)
//   mesh_solver.hip   the solver object: constraint generation, nested-dissection layout, launch sequence, C-ABI
#pragma once
#include "lvk_hip_internal.hpp"

namespace lvkmesh {

#ifndef LVK_MESH_TB
#define LVK_MESH_TB 8
#endif
#ifndef LVK_MESH_WIN_WAVES
#define LVK_MESH_WIN_WAVES 3
#endif
#ifndef LVK_MESH_FWD_WAVE
#define LVK_MESH_FWD_WAVE (LVK_MESH_WIN_WAVES + 1)
#endif
// k_mesh_solve's wavefronts by role: wavefront 0 walks the pivot chain, wavefront MS_FWD_WAVE carries the forward substitution and stores
// the columns of L, the others update the window.  (Wavefronts go to the CU's four SIMDs round robin: with three window wavefronts the two
// light roles share SIMD 0 and every window wavefront has a SIMD of its own.)
constexpr int MS_WIN_WAVES = LVK_MESH_WIN_WAVES, MS_FWD_WAVE = LVK_MESH_FWD_WAVE;
constexpr int MS_NT = 64 * (MS_WIN_WAVES + 2);
constexpr int MS_BULK = 64 * MS_WIN_WAVES;  // window threads
constexpr int MS_CA = 4, MS_TB = LVK_MESH_TB;   // register tile of a window thread: columns x band offsets
constexpr int MS_HB_MAX = 103;              // widest band phase 1 holds in registers (meshes up to 16 columns)
constexpr int MS_WP_MAX = 108;              // window columns: hb + 1 + (MS_CA - 1), rounded up to a multiple of MS_CA
constexpr int MS_PAD = 8;                   // zeros in front of the LDS columns (negative relative indices of the pivot's own group)
constexpr int MS_LCOL = MS_PAD + 2 * MS_WP_MAX + 2 * MS_TB + 8;
// LDS layouts.  All window wavefronts read their operands from the pivot column every step -- 15 values per thread, 23 KB per step
// through the CU's one LDS pipeline -- at addresses 4 m + MS_TB b + j (m: column group of the tile, b: its band).  In a plain array the
// 64 lanes of a read fall on a few banks (strides of 4 and 8 doubles over 32 double-wide banks): measured, every read took four passes
// and the LDS pipeline, not the arithmetic, set the pace of the factorisation.  So the pivot columns are stored with one spare slot
// after every four entries -- logical 4 y + r at 5 y + r: tiles with different (m + MS_TB / 4 b) hit different banks, equal ones the
// same address (a broadcast); MS_TB is a multiple of 4 so that r is a compile-time constant of every read.  The hand-over arrays are
// skewed the same way (one spare slot per band).
static_assert(MS_TB % 4 == 0 && MS_PAD % 4 == 0, "the padded LDS layout needs band boundaries at multiples of 4");
constexpr int ms_px(int x) { return 5 * ((x + 64) / 4 - 16) + (x + 64) % 4; }      // padded position of logical index x (x >= -64)
constexpr int MS_LCOL_P = ms_px(MS_LCOL) + 8;
constexpr int MS_COL_P = 128 + 128 / MS_TB + 1;                                     // col[]: logical t at t + t / MS_TB
constexpr int MS_NEXT_BAND = MS_CA * MS_TB + 1;                                     // next[]: one spare slot per band
constexpr double MS_Q = 4294967296.0;       // Q32
constexpr int MS_N_MAX = 2048;              // unknowns whose right-hand side fits the workgroup's LDS (16 x 64 vertices)
constexpr int MS_CHUNK = 16;                // rows of L staged per round of the backward substitution
constexpr int MS_RING = 192, MS_RPITCH = 196;   // a staged row: L(j, k) at slot k mod 192 (three 64-row blocks cover the band); pitch: rows 4 slots apart in the banks
static_assert(MS_HB_MAX + 1 <= 128, "three 64-row blocks cover the rows a pivot row reaches");
constexpr int MS_BANDS = (MS_HB_MAX + MS_TB) / MS_TB;                      // bands of MS_TB band offsets
constexpr int MS_PF = (MS_BANDS * MS_CA * MS_TB + 63) / 64;                // prefetched entries per lane of the forward wavefront and column group
constexpr int ms_tiles(int hb) { int t = 0; for (int b = 0; b < (hb + MS_TB) / MS_TB; b++) t += (hb - MS_TB * b + 3) / MS_CA; return t; }
static_assert(ms_tiles(MS_HB_MAX) <= MS_BULK, "one register tile per window thread");
static_assert(MS_BANDS * MS_TB <= 128 && MS_HB_MAX < 128, "a column is two registers per lane of the chain");

// One block of the nested dissection (oracle S5'): a band system of its own -- the block's vertex rows, then its separators --
// of which only the first n_elim pivots are eliminated.
struct MeshBlockDev
{
    int n, hb, n_elim, nbands;
    double* N; double* g0; double* wz; double* Lc; double* Rc; double* T;      // Rc: unscaled pivot columns (layout of Lc); T: trailing (n - n_elim)^2 window, row major
    const int* xs_of;                       // separator position n_elim + q -> index into the separator system's solution
    const int* nat_of;                      // own position -> natural unknown index
};

struct MeshArgs
{
    // ---- nested dissection (nd != 0): kernels launched with one workgroup per block take n, hb, ... and the arrays from blocks[blockIdx.x]
    int nd, nblocks, n_elim, n_nat;         // n_elim: pivots to eliminate (n for a whole system); n_nat: unknowns of the whole mesh
    const MeshBlockDev* blocks;
    double* Rc; double* T;                  // (per block, see MeshBlockDev)
    const int* ndst; const int* gdst;       // k_mesh_prepare: natural band entry / unknown -> position in the blocks' arrays (or -1)
    const int* ssrc; const int* gsrc; int s_entries, ns;      // k_nd_sep_assemble: separator band entry / unknown -> its (<= 2) sources in T / wz of the blocks
    const double* Tall; const double* wzall;
    double* xs;                             // separator solution (binary64)
    const double* sep_wz; const double* sep_Lc; int sep_hb, fuse_sep;      // k_mesh_backsolve over the blocks: the separator system's backward substitution runs inside (fuse_sep)
    double* X;                              // solution in natural order (binary64), gathered from the blocks
    const int* sep_nat;                     // separator unknown -> natural index
    unsigned* ticket;                       // last-block-done counter of the block-parallel kernels
    int cols, rows, n, hb, nbands;          // nbands: bands of MS_TB band offsets covering 0 .. hb
    const double* stat;                     // static band, column layout: entry (i, k), k <= i <= k + hb, at [k * (hb + 1) + (i - k)]
    long long* Nq; long long* gq;           // Q32 sums of the feature rows (same layout as stat / one per unknown)
    double* N; double* g0;                  // the assembled system (k_mesh_prepare)
    double* wz;                             // D^-1 L^-1 g, from k_mesh_solve to k_mesh_backsolve
    float* mesh;                            // previous solution (absolute tracking-frame coordinates), updated on success
    double* Lc;                             // columns of L: L(i, k) at [k * (hb + 1) + (i - k)]
    int* fidx; float* fw;                   // per feature: the 4 unknown indices (x components) and barycentric weights
    const float2* p1; const float2* p2;     // tracked / matched points
    const int* count; int n_pts;            // number of pairs: *count when count != nullptr (decided on the GPU), else n_pts
    int min_samples;
    float region_w, region_h, ts_gen, ts_now, threshold;
    int* flags;                             // device: bit 0 = a feature fell outside the mesh
    float* out_offsets; uint8_t* out_mask; int* out_status;      // device-visible host memory
};

__host__ __device__ __forceinline__ int band_groups(int hb, int b) { return (hb - MS_TB * b + 3) / MS_CA; }
constexpr int NT_TILE = 16, NT_JMAX = MS_HB_MAX + NT_TILE;
constexpr int MB_NT = 64 + 512;
constexpr int MG_NT = 1024;
constexpr int MG_HB_MAX = 1023;             // LDS copies of the pivot column (2 x 8 KB); motion_resolution up to 167 columns

// kernel launches of mesh.hip (its kernels live in an anonymous namespace, compiled with the unit's own flags: csrc/Makefile MESH_FLAGS)
void launch_assemble(unsigned blocks, hipStream_t stream, const MeshArgs& a);
void launch_prepare(hipStream_t stream, const MeshArgs& a);
void launch_solve(unsigned blocks, hipStream_t stream, const MeshArgs& a);
void launch_backsolve(unsigned blocks, hipStream_t stream, const MeshArgs& a);
void launch_sep_assemble(unsigned blocks, hipStream_t stream, const MeshArgs& a);
void launch_sep_assemble_tiled(unsigned blocks, hipStream_t stream, const MeshArgs& a, int tiles_per_col);
void launch_solve_generic(hipStream_t stream, const MeshArgs& a);
void launch_backsolve_generic(hipStream_t stream, const MeshArgs& a);

} // namespace lvkmesh
