// mg_stream_impl.h -- temporally blocked, wave-streaming Jacobi smoother for CDNA4.
//
// doSmoothing (src/MG_solver_CPU.cpp:573-625) runs `step` Jacobi sweeps and then a
// residual-type norm; the driver follows it with getResidual (:268).  Launched sweep by
// sweep that is 24 B of HBM traffic per point and sweep.  This kernel advances S sweeps
// (and, optionally, the residual / error stage) in ONE pass over HBM:
//
//   * a wave64 owns a vertical strip 64*COLS columns wide and marches down the rows of
//     its chunk; lane i holds COLS adjacent columns, so each row is ONE coalesced
//     16 B/lane (COLS = 2) load of U and one of F -- 1 KiB per wave instruction;
//   * time level l of row y needs level l-1 of rows y-1, y, y+1: the wave keeps a
//     two-row history per level in registers and, when input row y arrives, computes
//     level 1 of row y-1, level 2 of row y-2, ... level S of row y-S and finally the
//     residual of row y-S-1 (a software pipeline skewed by one row per level);
//   * the east/west neighbours that live in the adjacent lane come through DPP
//     wave shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1) -- no LDS, no barriers, waves
//     never synchronise;
//   * strips overlap by H = S+1 columns (rounded up to even) and chunks by S+1 rows on
//     each side: the halo is recomputed redundantly instead of exchanged (the overlap
//     re-reads hit L2: the block index is remapped so neighbouring tiles share an XCD);
//   * the next PF rows are always in flight (register FIFO of 2*PF slots, loop unrolled by
//     2*PF; every load is unconditional -- out-of-window rows/columns are clamped -- so the
//     compiler can count the loads in flight and waits with vmcnt(N), not vmcnt(0)).
//
// Two more stages can be fused into the same pass (COLS = 2 builds):
//   * IN_PROLONG: level 0 of a row is U_in + doProlongation(coarse) (:354 + :368),
//     evaluated while the row streams in (coarse rows are cached in registers and
//     rotate as the owner row advances; 1-D weights come from the host tables);
//   * RESTRICT: the signed residual rows are restricted on the fly into the next
//     level's F (:287) -- the lane whose column pair holds lo[cc] combines its pair (or
//     its right neighbour's, via one DPP shift) of two consecutive residual rows; the
//     residual D itself then never touches HBM.
// With both, a V-cycle level costs  F + U + F_coarse (18 B/point) on the way down and
// U + coarse + F + U (26 B/point) on the way up.
//
// Every point is updated with exactly the reference's expression and association order
// (-ffp-contract=off), so the result is bit-identical to S separate sweeps; a halo point
// computed twice gets the same bits twice.  Algorithmic traffic of S sweeps + residual:
// read U, read F, write U, write D = 32 B per point instead of 24*S + 24 + 8.
//
// This file is the kernel source for BOTH field types: it is included once with MG_REAL = double
// (mg_stream.hip, the fp64 path whose results are bit-identical to the reference) and once with
// MG_REAL = float (mg_stream_f32.hip, the fp32 smoother of the mixed-precision mode).  With
// MG_REAL = double every expression is exactly what it was before the split.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "mg_divconst.h"
#include "mg_internal.h"

#if !defined(MG_REAL) || !defined(MG_REAL_NS)
#error "define MG_REAL (double|float) and MG_REAL_NS (f64|f32) before including mg_stream_impl.h"
#endif

namespace mg {
namespace k {
namespace MG_REAL_NS {

typedef MG_REAL real_t;

// host-built transfer tables in the field type (weights) -- see mg_tables.cpp
struct StreamTables {
    const int *p_orow = nullptr, *p_ocol = nullptr;
    const real_t *p_rhi = nullptr, *p_rlo = nullptr, *p_chi = nullptr, *p_clo = nullptr;
    real_t c_dx = 0, c_dx_rcp = 0;
    const int *r_inv = nullptr;
    const real_t *r_w = nullptr, *r_wf = nullptr;
};


constexpr int PF_DEFAULT = 2;    // rows of U and F in flight per lane (FIFO of 2*PF slots)
constexpr int WAVES_PER_WG = 4;  // 4 adjacent strips of one chunk
constexpr int MAX_S = 4;

enum InMode { IN_LOAD = 0, IN_ZERO = 1, IN_PROLONG = 2 };

typedef real_t real2_t __attribute__((ext_vector_type(2)));

struct StreamParams {
    int N;
    real_t dx2, inv;
    const real_t *in;   // IN_ZERO: unused
    const real_t *F;
    real_t *out;
    real_t *D;          // nullptr: residual not stored
    int d_sign;
    double *part;       // nullptr: no error norm; else one partial per wave
    int rows_per_chunk;
    int groups;         // workgroups per chunk row
    int n_blocks;       // chunks * groups
    // row window (1-D row-slab decomposition): the fine arrays in/F/out/D start at global
    // row row_base and hold rows_local rows; this launch updates rows [own_y0, own_y1).
    // Single GPU: row_base = 0, rows_local = N, own = [0, N).
    int row_base, rows_local, own_y0, own_y1;
    int norm_y0, norm_y1;           // rows counted in the error norm (a subset of the updated rows)
    double *out_wide;               // fp32 fields only: store the result as fp64 here instead of `out`
    int coarse_base, coarse_rows;   // IN_PROLONG: window of the coarse array
    int raw_norm;                   // error output is the raw sum over the owned rows
    int nt_min_n;                   // grids at least this large store U/D non-temporally
    int fc_base, fc_rows;           // RESTRICT: window of Fc (global row of its first local row, rows)
    // IN_PROLONG: coarse grid and the host-built tables of doProlongation
    const real_t *coarse;
    int Nc;
    const int *p_orow, *p_ocol;
    const real_t *p_rhi, *p_rlo, *p_chi, *p_clo;
    real_t c_dx, c_dx_rcp;   // coarse spacing and RN(1/c_dx)
    // RESTRICT: next level's F and the host-built tables of doRestriction
    real_t *Fc;
    int M;
    const int *r_inv;   // [N] fine index -> coarse index whose lower-left sample it is, or -1
    const real_t *r_w;  // [M] weights by coarse index
    const real_t *r_wf; // [N] r_w[r_inv[x]] by fine index (0 where r_inv < 0)
};

// value of the neighbouring lane (lane-1 / lane+1); the lane at the wave edge reads 0
// (bound_ctrl: no previous destination value has to be materialised)
__device__ __forceinline__ double from_lane_below(double v)
{
    union { double d; int i[2]; } a, r;
    a.d = v;
    r.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
    r.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], 0x138, 0xf, 0xf, true);
    return r.d;
}
__device__ __forceinline__ double from_lane_above(double v)
{
    union { double d; int i[2]; } a, r;
    a.d = v;
    r.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
    r.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], 0x130, 0xf, 0xf, true);
    return r.d;
}
__device__ __forceinline__ float from_lane_below(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float from_lane_above(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}

// x / c for a constant c: div_by_const(x, c, RN(1/c)), see mg_divconst.h

// a - 4*u with one rounding: the product 4*u is exact in binary floating point, so the fused form is bit-identical
// to the reference's `... - 4*U` (src/MG_solver_CPU.cpp:590, :560) under -ffp-contract=off
__device__ __forceinline__ double minus4(double a, double u) { return __builtin_fma(-4.0, u, a); }
__device__ __forceinline__ float minus4(float a, float u) { return __builtin_fmaf(-4.0f, u, a); }

// Read-only host-built tables are read through the constant address space: the compiler may
// then use scalar loads (s_load, SGPR result, lgkmcnt) for wave-uniform indices instead of
// vector loads that would queue behind the streaming loads in vmcnt order.
typedef const int __attribute__((address_space(4))) *const_int_ptr;
typedef const real_t __attribute__((address_space(4))) *const_real_ptr;
__device__ __forceinline__ int table_i(const int *t, int i) { return ((const_int_ptr)(uintptr_t)t)[i]; }
__device__ __forceinline__ real_t table_d(const real_t *t, int i) { return ((const_real_ptr)(uintptr_t)t)[i]; }

template <int COLS>
struct Row {
    real_t v[COLS];
};

// Row loads are UNCONDITIONAL: rows and columns outside the window are clamped to a valid
// address instead of being skipped.  What such a load returns is never consumed by a point
// that is stored (a rim point keeps its value and never looks at its neighbours), and a load
// that is always issued lets the compiler count the loads in flight: with predicated loads it
// has to assume none was issued and waits for vmcnt(0) at every use.
template <int COLS>
__device__ __forceinline__ Row<COLS> load_row(const real_t *__restrict__ base)
{
    Row<COLS> r;
    if constexpr (COLS == 2) {
        const real2_t t = *reinterpret_cast<const real2_t *>(base);
        r.v[0] = t.x;
        r.v[1] = t.y;
    } else {
        r.v[0] = *base;
    }
    return r;
}

// nt: the array is far larger than the caches and is next read by a later kernel: a
// non-temporal store keeps it from displacing the halo rows and coarse rows the neighbouring
// tiles re-read (wave-uniform flag, set for N >= 2048)
template <int COLS>
__device__ __forceinline__ void store_row(real_t *__restrict__ base, const Row<COLS> &r, bool nt)
{
    if constexpr (COLS == 2) {
        real2_t t;
        t.x = r.v[0];
        t.y = r.v[1];
        if (nt) __builtin_nontemporal_store(t, reinterpret_cast<real2_t *>(base));
        else *reinterpret_cast<real2_t *>(base) = t;
    } else {
        *base = r.v[0];
    }
}

// halo columns per side: level S must be valid one column beyond the owned strip for the
// residual stage, two for the fused restriction (it reads the residual one column and one
// row beyond the owned tile); rounded up to even so strips stay 16 B aligned
template <int S, bool RESTRICT>
struct Halo {
    static constexpr int value = (S + (RESTRICT ? 2 : 1) + 1) & ~1;
};

// three consecutive coarse values of one coarse row, starting at this lane's base column
struct Coarse3 {
    real_t v[3];
};
__device__ __forceinline__ Coarse3 load_coarse(const real_t *__restrict__ coarse, int Nc, int base, int rows, int row,
                                               int col)
{
    // unconditional like load_row: rows fetched ahead of need may lie outside the local window
    // and columns past the grid are clamped (never consumed)
    Coarse3 c;
    int r = row - base;
    r = r < 0 ? 0 : (r < rows - 1 ? r : rows - 1);
    const real_t *b = coarse + (size_t)r * Nc;
    c.v[0] = b[col];
    c.v[1] = b[col + 1 < Nc ? col + 1 : Nc - 1];
    c.v[2] = b[col + 2 < Nc ? col + 2 : Nc - 1];
    return c;
}

template <int S, int COLS, int IN, bool RESTRICT, int PF = PF_DEFAULT>
__global__ __launch_bounds__(64 * WAVES_PER_WG) void k_jacobi_stream(const StreamParams p)
{
    static_assert(COLS == 2 || (IN != IN_PROLONG && !RESTRICT), "fused transfer stages need column pairs");
    constexpr int W = 64 * COLS;
    constexpr int H = Halo<S, RESTRICT>::value;
    constexpr int OW = W - 2 * H;  // columns a wave owns

    // blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous
    // range of tiles so halo re-reads hit its L2.  Speed only, never correctness.
    const int per_xcd = (p.n_blocks + 7) >> 3;
    const int tile = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (tile >= p.n_blocks) return;
    const int chunk = tile / p.groups;
    const int group = tile - chunk * p.groups;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int strip = group * WAVES_PER_WG + wave;
    const int N = p.N;
    const int own_x0 = strip * OW;
    if (own_x0 >= N) {  // no barriers in this kernel: a wave may leave at any time
        if (p.part && lane == 0) p.part[(size_t)tile * WAVES_PER_WG + wave] = 0.0;
        return;
    }
    const int y0 = p.own_y0 + chunk * p.rows_per_chunk;
    int y1 = y0 + p.rows_per_chunk;
    if (y1 > p.own_y1) y1 = p.own_y1;
    if (y0 >= p.own_y1) return;
    // rows that exist in the local window and in the grid
    const int av_lo = p.row_base > 0 ? p.row_base : 0;
    const int av_hi = p.row_base + p.rows_local < N ? p.row_base + p.rows_local : N;

    const int xl = own_x0 - H + lane * COLS;  // this lane's first column
    bool col_in[COLS], col_edge[COLS], col_even[COLS];
    bool lane_owns = true;
#pragma unroll
    for (int j = 0; j < COLS; ++j) {
        const int x = xl + j;
        col_in[j] = x >= 0 && x < N;
        col_edge[j] = x <= 0 || x >= N - 1;
        col_even[j] = (x & 1) == 0;
        lane_owns = lane_owns && x >= own_x0 && x < own_x0 + OW && x < N;
    }
    const bool lane_loads = col_in[0] && col_in[COLS - 1];  // COLS == 2: N even, xl even
    const real_t dx2 = p.dx2, inv = p.inv;
    const bool nt_stores = N >= p.nt_min_n;
    const bool want_res = RESTRICT || p.D != nullptr || p.part != nullptr;

    // ---- fused prolongation input: per-lane column tables, coarse row cache ----------
    int pc_base = 0;              // coarse column of this lane's first fine column
    bool pc_second_shift = false; // second fine column belongs to the next coarse cell
    real_t pc_hi[2] = {0.0, 0.0}, pc_lo[2] = {0.0, 0.0};
    Coarse3 c_lo = {{0.0, 0.0, 0.0}}, c_hi = {{0.0, 0.0, 0.0}};
    int c_row = -0x40000000;      // coarse row held in c_lo (wave-uniform); c_hi holds c_row + 1
    if constexpr (IN == IN_PROLONG) {
        if (lane_loads) {
            pc_base = p.p_ocol[xl];
            pc_second_shift = p.p_ocol[xl + 1] != pc_base;
            pc_hi[0] = p.p_chi[xl];
            pc_lo[0] = p.p_clo[xl];
            pc_hi[1] = p.p_chi[xl + 1];
            pc_lo[1] = p.p_clo[xl + 1];
        }
    }

    // ---- fused restriction output: which coarse column this lane produces -------------
    int rc_col = -1;          // coarse column (interior) or -1
    bool rc_shift = false;    // its lower-left fine sample is this lane's SECOND column
    real_t rw_a = 0.0, rw_b = 0.0;
    Row<COLS> d_prev;         // signed residual of the previous row
#pragma unroll
    for (int j = 0; j < COLS; ++j) d_prev.v[j] = 0.0;
    if constexpr (RESTRICT) {
        if (lane_owns) {
            const int ca = p.r_inv[xl], cb = p.r_inv[xl + 1];
            rc_col = ca >= 0 ? ca : cb;
            rc_shift = ca < 0 && cb >= 0;
            if (rc_col >= 0) {
                rw_a = p.r_w[rc_col];
                rw_b = real_t(1.0) - rw_a;  // src/MG_solver_CPU.cpp:665
            }
        }
    }
    // the rim of the next level's F is zero (doRestriction's memset, :651): rim columns are
    // written by the lanes that own fine columns 0 and N-1 alongside every coarse row, rim rows
    // by the chunks that hold fine rows 0 and N-1
    const bool first_col_lane = RESTRICT && lane_owns && xl == 0;
    const bool last_col_lane = RESTRICT && lane_owns && xl + COLS == N;
    if constexpr (RESTRICT) {
        for (int edge = 0; edge < 2; ++edge) {
            if (edge == 0 ? (y0 != 0) : (y1 != N)) continue;
            real_t *row = p.Fc + (size_t)((edge == 0 ? 0 : p.M - 1) - p.fc_base) * p.M;
            if (rc_col >= 0) row[rc_col] = 0.0;
            if (first_col_lane) row[0] = 0.0;
            if (last_col_lane) row[p.M - 1] = 0.0;
        }
    }

    // register state: two-row history per level, F delay line, prefetch FIFO
    Row<COLS> older[S + 1], newer[S + 1], fq[S + 2];
#pragma unroll
    for (int l = 0; l <= S; ++l)
#pragma unroll
        for (int j = 0; j < COLS; ++j) older[l].v[j] = newer[l].v[j] = 0.0;
#pragma unroll
    for (int l = 0; l <= S + 1; ++l)
#pragma unroll
        for (int j = 0; j < COLS; ++j) fq[l].v[j] = 0.0;

    const int y_first = y0 - (S + 1);                             // first input row
    const int T = (y1 - y0) + 2 * (S + 1) + (RESTRICT ? 1 : 0);   // input rows consumed
    const int y_end = y_first + T;                                // one past the last input row
    // clamped column of this lane's loads (lanes left/right of the grid re-read a valid pair)
    const size_t col_off = (size_t)(xl < 0 ? 0 : (xl > N - COLS ? N - COLS : xl));

    // wave-uniform per-row table entries travel through the same FIFO as the rows they
    // belong to, so their (scalar) loads are issued PF iterations before use
    // The FIFO has 2*PF slots and the loop body covers 2*PF rows: a slot is refilled PF rows
    // after it was consumed, so a load never targets a register whose old value is still live.
    // (With PF slots the compiler resolves the loop-carried slots by register copies at the
    // back edge, which read the in-flight loads and force s_waitcnt vmcnt(0) every PF rows.)
    constexpr int NB = 2 * PF;
    Row<COLS> pu[NB], pf[NB];
    Coarse3 pc[NB];               // IN_PROLONG: coarse row (owner + 1) of the input row, 3 columns
    int q_own[NB];                // IN_PROLONG: owner coarse row of the input row
    real_t q_yh[NB], q_yl[NB];    // IN_PROLONG: its two row weights
    int q_rc[NB];                 // RESTRICT: coarse row sampled at fine row (input row - S - 2)
    real_t q_rw[NB];              // RESTRICT: its weight c
#pragma unroll
    for (int k = PF; k < NB; ++k) {
#pragma unroll
        for (int j = 0; j < COLS; ++j) pu[k].v[j] = pf[k].v[j] = 0.0;
        pc[k].v[0] = pc[k].v[1] = pc[k].v[2] = 0.0;
        q_own[k] = q_rc[k] = -1;
        q_yh[k] = q_yl[k] = q_rw[k] = 0.0;
    }
#pragma unroll
    for (int k = 0; k < PF; ++k) {
        const int y = y_first + k;
        const bool row_ok = y >= av_lo && y < av_hi && y < y_end;
        const int yc = y < av_lo ? av_lo : (y < av_hi ? y : av_hi - 1);  // clamped into the window
        const size_t off = (size_t)(yc - p.row_base) * N + col_off;
        if constexpr (IN != IN_ZERO) pu[k] = load_row<COLS>(p.in + off);
        pf[k] = load_row<COLS>(p.F + off);
        q_own[k] = -1;
        q_yh[k] = q_yl[k] = q_rw[k] = 0.0;
        q_rc[k] = -1;
        pc[k].v[0] = pc[k].v[1] = pc[k].v[2] = 0.0;
        if constexpr (IN == IN_PROLONG) {
            if (row_ok) {
                q_own[k] = table_i(p.p_orow, y);
                q_yh[k] = table_d(p.p_rhi, y);
                q_yl[k] = table_d(p.p_rlo, y);
            }
            // the UPPER coarse row of this input row travels with it through the FIFO, so every
            // vector load of the loop is issued at a fixed place PF iterations before its use
            pc[k] = load_coarse(p.coarse, p.Nc, p.coarse_base, p.coarse_rows, q_own[k] + 1, pc_base);
        }
        if constexpr (RESTRICT) {
            const int yl = y - S - 2;
            if (yl >= y0 && yl < y1) {
                q_rc[k] = table_i(p.r_inv, yl);
                q_rw[k] = table_d(p.r_wf, yl);
            }
        }
    }

    if constexpr (IN == IN_PROLONG) {
        // before the first rotation c_hi must hold the owner row of the first input row
        const int ys = y_first > av_lo ? y_first : av_lo;
        if (ys < av_hi && ys < y_end) {
            const int i0 = table_i(p.p_orow, ys);
            c_hi = load_coarse(p.coarse, p.Nc, p.coarse_base, p.coarse_rows, i0, pc_base);
            c_row = i0 - 1;
        }
    }

    double acc = 0.0;

    for (int t0 = 0; t0 < T; t0 += NB) {
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int kr = (k + PF) % NB;   // the slot refilled while slot k is consumed
            const int yin = y_first + t0 + k;
            Row<COLS> nw, cf = pf[k];
            if constexpr (IN == IN_ZERO) {
#pragma unroll
                for (int j = 0; j < COLS; ++j) nw.v[j] = 0.0;
            } else {
                nw = pu[k];
            }
            const int own_i = q_own[k];
            const real_t own_yh = q_yh[k], own_yl = q_yl[k];
            const Coarse3 own_up = pc[k];
            const int rc_row = q_rc[k];
            const real_t rc_w = q_rw[k];
            {   // refill the slot consumed PF rows ago with the row PF ahead
                const int y = yin + PF;
                const bool row_ok = y >= av_lo && y < av_hi && y < y_end;
                const int yc = y < av_lo ? av_lo : (y < av_hi ? y : av_hi - 1);
                const size_t off = (size_t)(yc - p.row_base) * N + col_off;
                if constexpr (IN != IN_ZERO) pu[kr] = load_row<COLS>(p.in + off);
                pf[kr] = load_row<COLS>(p.F + off);
                if constexpr (IN == IN_PROLONG) {
                    q_own[kr] = -1;
                    if (row_ok) {
                        q_own[kr] = table_i(p.p_orow, y);
                        q_yh[kr] = table_d(p.p_rhi, y);
                        q_yl[kr] = table_d(p.p_rlo, y);
                    }
                    pc[kr] = load_coarse(p.coarse, p.Nc, p.coarse_base, p.coarse_rows, q_own[kr] + 1, pc_base);
                }
                if constexpr (RESTRICT) {
                    const int yl = y - S - 2;
                    q_rc[kr] = -1;
                    if (yl >= y0 && yl < y1) {
                        q_rc[kr] = table_i(p.r_inv, yl);
                        q_rw[kr] = table_d(p.r_wf, yl);
                    }
                }
            }

            if constexpr (IN == IN_PROLONG) {
                // level 0 = U + P(coarse): doProlongation :700 as a gather, then
                // doGridAddition :569 (U1 = U1 + U2).  own_i is wave-uniform.
                if (own_i >= 0) {
                    if (own_i != c_row) {  // the owner row advanced by one (host-checked): rotate
                        c_lo = c_hi;
                        c_row = own_i;
                    }
                    c_hi = own_up;         // row own_i + 1, loaded PF iterations ago
                    const real_t c_dx = p.c_dx, c_rcp = p.c_dx_rcp;
#pragma unroll
                    for (int j = 0; j < COLS; ++j) {
                        const bool sh = j == 1 && pc_second_shift;
                        const real_t c1 = sh ? c_lo.v[1] : c_lo.v[0], c2 = sh ? c_lo.v[2] : c_lo.v[1];
                        const real_t c3 = sh ? c_hi.v[1] : c_hi.v[0], c4 = sh ? c_hi.v[2] : c_hi.v[1];
                        const real_t xh = pc_hi[j], xlo = pc_lo[j];
                        const real_t num = (c1 * xh + c2 * xlo) * own_yh + (c3 * xh + c4 * xlo) * own_yl;
                        const real_t pv = div_by_const(div_by_const(num, c_dx, c_rcp), c_dx, c_rcp);
                        nw.v[j] = nw.v[j] + pv;
                    }
                }
            }

#pragma unroll
            for (int l = S + 1; l >= 1; --l) fq[l] = fq[l - 1];
            fq[0] = cf;

            // levels 1..S: level l produces row yin-l from level l-1 rows yin-l-1, yin-l, yin-l+1
#pragma unroll
            for (int l = 1; l <= S; ++l) {
                const int y = yin - l;
                const bool row_edge = y <= 0 || y >= N - 1;
                const Row<COLS> c = newer[l - 1], so = older[l - 1];
                const real_t west0 = from_lane_below(c.v[COLS - 1]);
                const real_t east_last = from_lane_above(c.v[0]);
                Row<COLS> o;
#pragma unroll
                for (int j = 0; j < COLS; ++j) {
                    const real_t w = j == 0 ? west0 : c.v[j > 0 ? j - 1 : 0];
                    const real_t e = j == COLS - 1 ? east_last : c.v[j < COLS - 1 ? j + 1 : 0];
                    // src/MG_solver_CPU.cpp:590: U += 0.25*(U[i+1]+U[i-1]+U[j+1]+U[j-1] - 4U - dx^2 F)
                    // `- 4*U` through one fma: 4*U is exact, so fma(-4, U, a) is the same bits as a - 4*U (one VALU op less)
                    const real_t t = minus4(nw.v[j] + so.v[j] + e + w, c.v[j]) - dx2 * fq[l].v[j];
                    o.v[j] = c.v[j] + real_t(0.25) * t;
                }
                // the rim keeps its value: ONE select per value on the merged condition (the compiler turns the
                // two nested wave-uniform cases into two selects per value otherwise: 12 more VALU ops per row)
#pragma unroll
                for (int j = 0; j < COLS; ++j) o.v[j] = (row_edge || col_edge[j]) ? c.v[j] : o.v[j];
                older[l - 1] = c;
                newer[l - 1] = nw;
                nw = o;
            }

            // nw is level S of row yin-S: the smoothed U
            {
                const int y = yin - S;
                if (y >= y0 && y < y1 && lane_owns) {
                    if (sizeof(real_t) != sizeof(double) && p.out_wide) {
                        // mixed precision: the last node of a cycle hands its result over in fp64 (exact
                        // widening) instead of leaving it to a separate conversion pass
                        double *w = p.out_wide + (size_t)(y - p.row_base) * N + xl;
#pragma unroll
                        for (int j = 0; j < COLS; ++j) __builtin_nontemporal_store((double)nw.v[j], w + j);
                    } else {
                        store_row<COLS>(p.out + (size_t)(y - p.row_base) * N + xl, nw, nt_stores);
                    }
                }
            }

            // residual stage, row yin-S-1 (src/MG_solver_CPU.cpp:560 and the error sums :611)
            if (want_res) {
                const int y = yin - S - 1;
                const bool mine = y >= y0 && y < y1 && lane_owns;  // each point counted once
                const bool counted = mine && y >= p.norm_y0 && y < p.norm_y1;
                const bool row_edge = y <= 0 || y >= N - 1;
                const Row<COLS> c = newer[S], so = older[S];
                const real_t west0 = from_lane_below(c.v[COLS - 1]);
                const real_t east_last = from_lane_above(c.v[0]);
                Row<COLS> d;
#pragma unroll
                for (int j = 0; j < COLS; ++j) {
                    const real_t w = j == 0 ? west0 : c.v[j > 0 ? j - 1 : 0];
                    const real_t e = j == COLS - 1 ? east_last : c.v[j < COLS - 1 ? j + 1 : 0];
                    const real_t r = inv * minus4(nw.v[j] + so.v[j] + e + w, c.v[j]) - fq[S + 1].v[j];
                    const bool interior = !(row_edge || col_edge[j]);
                    const real_t dv = interior ? r : real_t(0.0);
                    d.v[j] = p.d_sign < 0 ? -dv : dv;  // the driver's sign flip :277-280
                    // (row+col) even interior points only, :610/:617
                    if (counted && interior && (((y & 1) == 0) == col_even[j])) acc += fabs((double)r);  // norms are accumulated in fp64 whatever the field type
                }
                if (mine && p.D) store_row<COLS>(p.D + (size_t)(y - p.row_base) * N + xl, d, nt_stores);

                if constexpr (RESTRICT) {
                    // doRestriction :656-678 on rows (y-1, y) of the signed residual: coarse row
                    // rc has its lower-left sample in fine row y-1
                    // (rc_row / rc_w came through the FIFO: wave-uniform, -1 = no coarse row has its
                    // lower-left sample in fine row y-1 of this chunk)
                    {
                        if (rc_row >= 0) {
                            const real_t wc = rc_w, wd = real_t(1.0) - wc;  // c, d of :664-666
                            const real_t p_up = from_lane_above(d_prev.v[0]);
                            const real_t q_up = from_lane_above(d.v[0]);
                            const real_t u0 = rc_shift ? d_prev.v[1] : d_prev.v[0];
                            const real_t u1 = rc_shift ? p_up : d_prev.v[1];
                            const real_t u2 = rc_shift ? d.v[1] : d.v[0];
                            const real_t u3 = rc_shift ? q_up : d.v[1];
                            // :676  U_c = b*d*U_f[f] + a*d*U_f[f+1] + c*b*U_f[f+N] + a*c*U_f[f+N+1]
                            const real_t vc = rw_b * wd * u0 + rw_a * wd * u1 + wc * rw_b * u2 + rw_a * wc * u3;
                            real_t *crow = p.Fc + (size_t)(rc_row - p.fc_base) * p.M;
                            if (rc_col >= 0) crow[rc_col] = vc;
                            if (first_col_lane) crow[0] = 0.0;
                            if (last_col_lane) crow[p.M - 1] = 0.0;
                        }
                    }
                    d_prev = d;
                }
            }
            older[S] = newer[S];
            newer[S] = nw;
        }
    }
    if (p.part) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
        if (lane == 0) p.part[(size_t)tile * WAVES_PER_WG + wave] = acc;
    }
}

// One launch: tile the grid for ONE resident round of workgroups (measured occupancy of
// this instantiation x CUs) where the grid is large enough, never fewer than 8 rows per
// chunk (each chunk re-reads 2(S+1) halo rows), then the fixed-order error reduction.
template <int S, int COLS, int IN, bool RESTRICT, int PF = PF_DEFAULT>
void launch_k(hipStream_t s, StreamParams p, double *err_out)
{
    static int blocks_per_cu = 0;
    if (blocks_per_cu == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_jacobi_stream<S, COLS, IN, RESTRICT, PF>, 64 * WAVES_PER_WG, 0) != hipSuccess || n < 1) {
            (void)hipGetLastError();
            n = 2;
        }
        blocks_per_cu = n > 8 ? 8 : n;
    }
    const int N = p.N;
    const int own = p.own_y1 - p.own_y0;  // rows this launch updates (N on a single GPU)
    if (own <= 0) return;
    constexpr int OW = 64 * COLS - 2 * Halo<S, RESTRICT>::value;
    const int strips = (N + OW - 1) / OW;
    const int groups = (strips + WAVES_PER_WG - 1) / WAVES_PER_WG;
    static const int resident_pct = [] { const char *e = getenv("MG_RESIDENT_PCT"); return e ? atoi(e) : 100; }();
    const int resident = ctx().n_cu * blocks_per_cu * resident_pct / 100;
    int chunks = resident / groups;
    // small grids are latency bound on the length of a wave's march: shorter chunks, more waves
    static const int min_rows_forced = [] { const char *e = getenv("MG_MIN_ROWS"); return e ? atoi(e) : 0; }();
    const int min_rows = min_rows_forced ? min_rows_forced : (N <= 256 ? 2 : N <= 1024 ? 4 : 8);
    const int max_chunks = (own + min_rows - 1) / min_rows;
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks < 1) chunks = 1;
    int rows = (own + chunks - 1) / chunks;
    static const int rows_cap = [] { const char *e = getenv("MG_MAX_ROWS"); return e ? atoi(e) : 0; }();
    if (rows_cap > 0 && rows > rows_cap) rows = rows_cap;  // tuning knob: more, shorter tiles
    chunks = (own + rows - 1) / rows;
    p.rows_per_chunk = rows;
    p.groups = groups;
    p.n_blocks = chunks * groups;
    p.part = nullptr;
    const size_t n_part = (size_t)p.n_blocks * WAVES_PER_WG;
    if (err_out) {
        p.part = norm_partials(n_part);  // every wave of every tile writes its slot
        if (!p.part) return;
    }
    const int grid = ((p.n_blocks + 7) / 8) * 8;
    hipLaunchKernelGGL((k_jacobi_stream<S, COLS, IN, RESTRICT, PF>), dim3(grid), dim3(64 * WAVES_PER_WG), 0, s, p);
    // a slab launch leaves its RAW partial sum; the caller combines the slabs in rank order
    if (err_out) norm_finish(s, p.part, n_part, p.raw_norm ? -1 : N, err_out);
}

template <int S, int PF>
void launch_variant(hipStream_t s, const StreamParams &p, double *err_out)
{
    const bool restrict_out = p.Fc != nullptr, prolong_in = p.coarse != nullptr, zero = p.in == nullptr;
    if (p.N % 2 != 0) {  // 8 B lanes: odd row pitch; the fused transfer stages are not built for it
        if (zero) launch_k<S, 1, IN_ZERO, false, PF>(s, p, err_out);
        else launch_k<S, 1, IN_LOAD, false, PF>(s, p, err_out);
    } else if (restrict_out) {
        if (zero) launch_k<S, 2, IN_ZERO, true, PF>(s, p, err_out);
        else launch_k<S, 2, IN_LOAD, true, PF>(s, p, err_out);
    } else if (prolong_in) {
        launch_k<S, 2, IN_PROLONG, false, PF>(s, p, err_out);
    } else {
        if (zero) launch_k<S, 2, IN_ZERO, false, PF>(s, p, err_out);
        else launch_k<S, 2, IN_LOAD, false, PF>(s, p, err_out);
    }
}

template <int S>
void launch_steps(hipStream_t s, const StreamParams &p, double *err_out)
{
    // rows in flight per lane: PF = 2 (a FIFO of 4 slots).  A deeper FIFO (PF = 3 at the same occupancy, PF = 4, 8) costs registers
    // and a longer prologue and measured slower at every size from 128 to 8192 once all loads were
    // unconditional (fused prolongation: 18.1 vs 21.1 us at N = 1024, 10.9 vs 13.7 us at N = 128).
    launch_variant<S, 2>(s, p, err_out);
}


// entry point of this instantiation: same contract as k::jacobi_stream (mg_internal.h) in the
// field type real_t; tb carries the tables the fused stages need
inline void run(hipStream_t s, int N, real_t dx2, real_t inv, const real_t *in, const real_t *F, real_t *out, int steps,
                double *err_out, real_t *D_out, int d_sign, const real_t *coarse, int Nc, real_t *Fc, int M,
                const StreamTables &tb, const RowWindow *fine_w, const RowWindow *coarse_w, const RowWindow *fc_w,
                double *out_wide = nullptr)
{
    if (steps < 1 || steps > MAX_S) {
        fail(MG_ERR_ARG, "jacobi_stream: %d sweeps per launch (1..%d)", steps, MAX_S);
        return;
    }
    if ((coarse || Fc) && (N < 4 || N % 2 != 0)) {
        fail(MG_ERR_ARG, "jacobi_stream: fused transfer stages need an even grid size (N=%d)", N);
        return;
    }
    StreamParams p = {};
    p.N = N;
    p.dx2 = dx2;
    p.inv = inv;
    p.in = in;
    p.F = F;
    p.out = out;
    p.out_wide = out_wide;
    p.D = D_out;
    p.d_sign = d_sign;
    p.row_base = fine_w ? fine_w->base : 0;
    p.rows_local = fine_w ? fine_w->rows : N;
    p.own_y0 = fine_w ? fine_w->own_lo : 0;
    p.own_y1 = fine_w ? fine_w->own_hi : N;
    p.norm_y0 = fine_w && fine_w->norm_lo >= 0 ? fine_w->norm_lo : p.own_y0;
    p.norm_y1 = fine_w && fine_w->norm_lo >= 0 ? fine_w->norm_hi : p.own_y1;
    p.raw_norm = fine_w ? 1 : 0;
    static const int nt_min = [] { const char *e = getenv("MG_NT_MIN_N"); return e ? atoi(e) : 2048; }();
    p.nt_min_n = nt_min;
    if (coarse) {
        p.coarse_base = coarse_w ? coarse_w->base : 0;
        p.coarse_rows = coarse_w ? coarse_w->rows : Nc;
        p.coarse = coarse;
        p.Nc = Nc;
        p.p_orow = tb.p_orow;
        p.p_ocol = tb.p_ocol;
        p.p_rhi = tb.p_rhi;
        p.p_rlo = tb.p_rlo;
        p.p_chi = tb.p_chi;
        p.p_clo = tb.p_clo;
        p.c_dx = tb.c_dx;
        p.c_dx_rcp = tb.c_dx_rcp;
    }
    if (Fc) {
        p.fc_base = fc_w ? fc_w->base : 0;
        p.fc_rows = fc_w ? fc_w->rows : M;
        p.Fc = Fc;
        p.M = M;
        p.r_inv = tb.r_inv;
        p.r_w = tb.r_w;
        p.r_wf = tb.r_wf;
    }
    switch (steps) {
        case 1: launch_steps<1>(s, p, err_out); break;
        case 2: launch_steps<2>(s, p, err_out); break;
        case 3: launch_steps<3>(s, p, err_out); break;
        default: launch_steps<4>(s, p, err_out); break;
    }
}

}  // namespace MG_REAL_NS
}  // namespace k
}  // namespace mg
