// mg_tile_impl.h -- the fused node kernels of the SMALL levels (64 < N <= ~1024): a 2-D tile held in registers.
//
// The wave-streaming smoother (mg_stream_impl.h) is built for levels that do not fit the caches: a wave marches down
// its column strip row by row through a software pipeline.  On a small level that march IS the launch: a dozen rows
// of pipeline fill and drain at ~0.44 us per row step, a prologue of two dependent memory round trips, ~10 us per
// launch whatever the bytes (profiles/r02_vcycle_levels.txt: 8 launches of 9.7-17.4 us for the levels 128...1024 of a
// V-cycle, 240 of them in a W-cycle).  Here the same node is laid out the other way round:
//
//   * a workgroup owns a tile of TY x TX points and holds the window own +- HALO (HALO = S + 1 rows and columns: the
//     halo is recomputed redundantly, never exchanged between workgroups) ENTIRELY IN REGISTERS: lane = column (64
//     columns per window), wave w = RPW consecutive rows, one value per lane and row;
//   * a sweep updates all rows of the window at once: the north/south neighbours of a row are the wave's own
//     registers, except for the first and the last row of its block, which come from the neighbouring waves through
//     two rows of LDS (double-buffered: one barrier per sweep); east/west neighbours are the adjacent lanes (DPP
//     wave_shr/shl:1, as in the streaming kernel);
//   * all loads of the node (F, U, the coarse rows of the fused prolongation, the per-row and per-column transfer
//     tables) are issued up front in one batch -- one memory round trip, then S barrier-separated sweeps of a few
//     hundred cycles each;
//   * the fused input and output stages are those of the streaming kernel: level 0 = zero | U | U + P(coarse)
//     (src/MG_solver_CPU.cpp:256 | - | :354 + :368), then the sweeps (:587-599), the error norm (:607-622), and
//     optionally the signed residual restricted into the next level's F (:268, :277-280, :287).
//
// Every point is evaluated with the expressions of the streaming kernel (mg_lane_ops.h, mg_divconst.h: the reference's
// association order under -ffp-contract=off), so the arrays are bit-identical to it and to the reference; a halo
// point computed by two tiles gets the same bits twice.  Norm partials are summed in another order (per wave, then
// the fixed-order finish of mg_kernels.hip): the scalar agrees to rounding, as between any two kernels here.
//
// Kernel source for both field types (MG_REAL = double: mg_tile.hip; float: mg_tile_f32.hip).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "mg_divconst.h"
#include "mg_internal.h"
#include "mg_lane_ops.h"

#if !defined(MG_REAL) || !defined(MG_REAL_NS)
#error "define MG_REAL (double|float) and MG_REAL_NS (f64|f32) before including mg_tile_impl.h"
#endif

namespace mg {
namespace k {
namespace MG_REAL_NS {
namespace tile {

typedef MG_REAL real_t;

enum InMode { T_ZERO = 0, T_LOAD = 1, T_PROLONG = 2 };

struct TileParams {
    int N;
    real_t dx2, inv;
    const real_t *in;   // T_ZERO: unused
    const real_t *F;
    real_t *out;
    int no_out;         // the smoothed U is not stored
    int d_sign;
    double *part;       // nullptr: no error norm; else one partial per wave
    long long *trace;   // MG_TILE_TRACE builds only
    int tiles_x, n_blocks;
    // row window (1-D row-slab decomposition): the fine arrays in/F/out start at global row row_base and hold rows_local
    // rows; this launch updates rows [own_y0, own_y1), of which [norm_y0, norm_y1) count in the error norm.  The coarse
    // input and the coarse output have windows of their own.  Single GPU: row_base = 0, rows_local = N, own = [0, N).
    int row_base, rows_local, own_y0, own_y1, norm_y0, norm_y1;
    int coarse_base, coarse_rows, fc_base;
    // T_PROLONG: coarse grid and the host-built tables of doProlongation
    const real_t *coarse;
    int Nc;
    const int *p_orow, *p_ocol;
    const real_t *p_rhi, *p_rlo, *p_chi, *p_clo;
    real_t c_dx, c_dx_rcp;
    int own_closed;     // ProlongTable::closed_form: owner(k) = min(k*(Nc-1)/(N-1), Nc-2), formed in the kernel
    float own_rcp;      // 1/(N-1), for that division
    // RESTRICT: next level's F and the host-built tables of doRestriction
    real_t *Fc;
    int M;
    const int *r_inv;   // [N] fine index -> interior coarse index whose lower-left sample it is, or -1
    const real_t *r_w;  // [M] weights by coarse index
    const real_t *r_wf; // [N] r_w[r_inv[x]] by fine index (0 where r_inv < 0)
    // a batch of instances of the node (mg_internal.h: NodeBatch): blockIdx.y picks the instance's arrays
    const NodeBatchItem *batch;
    int part_stride;            // partials per instance
    int n_batch;                // (host side only)
    double *const *err_outs;
};

// window rows/columns beyond the owned tile, per side.  Level s of a point is valid when the point lies at least s rows
// and columns inside the window (every sweep loses one ring; from the zero field level 1 needs no neighbours, so one
// ring less is lost).  The error norm needs the residual on the owned tile (level S one ring beyond it), the fused
// restriction the residual one row and column beyond the tile (level S two rings beyond it).
template <int S, int IN, bool RESTRICT>
struct Geom {
    static constexpr int HALO = S + 1 + ((RESTRICT && IN != T_ZERO) ? 1 : 0);
    static constexpr int TX = 64 - 2 * HALO;
};

template <int S, int IN, bool RESTRICT, int RPW, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_jacobi_tile(const TileParams p)
{
    constexpr int HALO = Geom<S, IN, RESTRICT>::HALO;
    constexpr int RH = RPW * WAVES;       // rows of the window
    constexpr int TY = RH - 2 * HALO;     // rows the tile owns
    constexpr int TX = Geom<S, IN, RESTRICT>::TX;
    static_assert(TY >= 2 && TX >= 2 && RPW >= 2 && RPW <= 32, "window too small for its halo");
    // first and last row of every wave's block, of the level being swept; [parity]: the next exchange writes the other
    // half while a slow wave may still read this one (one barrier per exchange)
    __shared__ real_t xch[2][WAVES][2][64];

    // blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous range of tiles (a band of rows),
    // so the halo re-reads of neighbouring tiles hit its L2.  Speed only, never correctness.
#ifdef MG_TILE_TRACE   // diagnostics: where a launch spends its time (100 MHz timestamps of the middle tile's wave 0)
    const long long tr0 = wall_clock64();
#define TILE_STAMP(k) do { if (p.trace && tile_id == p.n_blocks / 2 && threadIdx.x == 0) p.trace[k] = wall_clock64(); } while (0)
#else
#define TILE_STAMP(k) do { } while (0)
#endif
    const int per_xcd = (p.n_blocks + 7) >> 3;
    const int tile_id = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (tile_id >= p.n_blocks) return;   // (the whole workgroup: no barrier is left waiting)
    // the arrays of this instance (wave-uniform; a batch reads them through the constant address space: scalar loads)
    const real_t *a_in = p.in, *a_F = p.F, *a_coarse = p.coarse;
    real_t *a_out = p.out, *a_Fc = p.Fc;
    double *a_part = p.part;
    if (p.batch) {
        typedef const NodeBatchItem __attribute__((address_space(4))) *item_ptr;
        const item_ptr b = (item_ptr)(uintptr_t)(p.batch + blockIdx.y);
        a_in = static_cast<const real_t *>(b->in);
        a_F = static_cast<const real_t *>(b->F);
        a_coarse = static_cast<const real_t *>(b->coarse);
        a_out = static_cast<real_t *>(b->out);
        a_Fc = static_cast<real_t *>(b->Fc);
        if (a_part) a_part += (size_t)blockIdx.y * (size_t)p.part_stride;
    }
    const int tile_y = tile_id / p.tiles_x, tile_x = tile_id - tile_y * p.tiles_x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int N = p.N;
    const int oy0 = p.own_y0 + tile_y * TY, oy1 = oy0 + TY < p.own_y1 ? oy0 + TY : p.own_y1;
    // rows that exist in the local window and in the grid (loads are clamped to them; a slab's window holds its rows
    // +- the halo its schedule promised, which is what this kernel reads: own +- HALO)
    const int av_lo = p.row_base > 0 ? p.row_base : 0;
    const int av_hi = p.row_base + p.rows_local < N ? p.row_base + p.rows_local : N;
    const int ox0 = tile_x * TX, ox1 = ox0 + TX < N ? ox0 + TX : N;
    const int x = ox0 - HALO + lane;        // this lane's column
    const int yb = oy0 - HALO + wave * RPW; // first row of this wave's block

    // ---- column state (lane constants)
    const bool col_edge = x <= 0 || x >= N - 1;       // rim column, or outside the grid
    const bool lane_owns = x >= ox0 && x < ox1;
    const int xc = x < 0 ? 0 : (x < N ? x : N - 1);   // loads are unconditional: columns/rows outside the grid are clamped to a
                                                      // valid address; what they return never reaches a stored point
    // rim points keep their value: the update is U + q*t with q = 0.25 inside and 0 on the rim (row: AND with a mask)
    const real_t qc = col_edge ? real_t(0.0) : real_t(0.25);
    const real_t dx2 = p.dx2, inv = p.inv;
    const bool want_res = RESTRICT || p.part != nullptr;

    auto row_y = [&](int j) { return yb + j; };
    auto row_clamped = [&](int j) { const int y = yb + j; return (y < av_lo ? av_lo : (y < av_hi ? y : av_hi - 1)) - p.row_base; };  // LOCAL row
    auto row_inner = [&](int j) { return ((unsigned)(yb + j - 1) < (unsigned)(N - 2)) ? -1 : 0; };  // 0 on the rim rows and outside
    auto row_owned = [&](int j) { return (unsigned)(yb + j - oy0) < (unsigned)(oy1 - oy0); };

    // ---- every load of the node, issued in one batch ---------------------------------------------------------------
    // wave-uniform per-row table entries ride in one register per table, row j of the block in lane j (v_readlane)
    const int yl = yb + (lane < RPW ? lane : RPW - 1);
    const int ylc = yl < 0 ? 0 : (yl < N ? yl : N - 1);   // (tables are indexed by the GLOBAL row)
    // The owner cells of the fused prolongation are ADDRESS ingredients of the coarse loads: read from their tables they
    // put a memory round trip in front of those loads.  For the level pairs of a halving hierarchy the owner is
    // min(k*(Nc-1)/(N-1), Nc-2) in integer arithmetic -- the host checked that against the reference's ceil() tables
    // entry by entry (ProlongTable::closed_form) -- so the kernel forms it itself; other pairs read the tables, first of
    // all loads.  (Measured: the 1.2-4 us from a workgroup's start to its last load issued did NOT shrink -- they are
    // kernel arguments, instruction fetch and, at N = 1024, 62 loads per lane through the texture addressers; the closed
    // form stays because it removes a dependent load, MG_TILE_NO_CLOSED_FORM=1 is the A/B switch.)
    int t_own = 0, cj = 0;
    if constexpr (IN == T_PROLONG) {
        if (p.own_closed) {
            auto owner_of = [&](int k) {   // k*(Nc-1) < 2^23: exact in fp32; the quotient estimate is off by at most one
                const int n = k * (p.Nc - 1), d = N - 1;
                int q = (int)((float)n * p.own_rcp);
                const int r = n - q * d;
                q += (r >= d) ? 1 : 0;
                q -= (r < 0) ? 1 : 0;
                return q < p.Nc - 2 ? q : p.Nc - 2;
            };
            t_own = owner_of(ylc);
            cj = owner_of(xc);
        } else {
            t_own = p.p_orow[ylc];
            cj = p.p_ocol[xc];
        }
    }
    real_t f[RPW], v[RPW];
#pragma unroll
    for (int j = 0; j < RPW; ++j) f[j] = a_F[(size_t)row_clamped(j) * N + xc];
    if constexpr (IN != T_ZERO) {
#pragma unroll
        for (int j = 0; j < RPW; ++j) v[j] = a_in[(size_t)row_clamped(j) * N + xc];
    }
    // local row of the coarse window (rows fetched for halo rows of the fine window may lie outside it: clamped, never consumed)
    auto coarse_local = [&](int r) { const int l = r - p.coarse_base; return l < 0 ? 0 : (l < p.coarse_rows - 1 ? l : p.coarse_rows - 1); };
    real_t t_rhi = 0, t_rlo = 0, pc_hi = 0, pc_lo = 0;
    real_t ca[RPW], cb[RPW], c0a = 0, c0b = 0;
    int own[RPW];
    if constexpr (IN == T_PROLONG) {
        t_rhi = p.p_rhi[ylc];
        t_rlo = p.p_rlo[ylc];
        pc_hi = p.p_chi[xc];
        pc_lo = p.p_clo[xc];
        const int last = p.Nc - 1;
        const int cj1 = cj + 1 < last ? cj + 1 : last;
        // coarse rows owner and owner + 1 of every fine row (the owner advances by at most one per fine row, host-checked:
        // ProlongTable::fusable): row (owner + 1) travels with its fine row, row owner(first row) once
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            own[j] = lane_value(t_own, j);
            const int up = own[j] + 1 < last ? own[j] + 1 : last;
            const real_t *crow = a_coarse + (size_t)coarse_local(up) * p.Nc;
            ca[j] = crow[cj];
            cb[j] = crow[cj1];
        }
        const real_t *crow0 = a_coarse + (size_t)coarse_local(own[0]) * p.Nc;
        c0a = crow0[cj];
        c0b = crow0[cj1];
    }
    int t_rc = -1, rc_col = -1;
    real_t t_rw = 0, rw_a = 0, rw_b = 0;
    if constexpr (RESTRICT) {
        const bool ok = yl >= oy0 && yl < oy1 && lane < RPW;   // rows outside the tile sample nothing
        const int rc = p.r_inv[ylc];
        t_rw = p.r_wf[ylc];
        t_rc = ok ? rc : -1;
        if (lane_owns) {
            rc_col = p.r_inv[x];
            rw_a = p.r_wf[x];           // = r_w[rc_col], by fine index: no load behind a load
            rw_b = real_t(1.0) - rw_a;  // src/MG_solver_CPU.cpp:665
        }
    }

#ifdef MG_TILE_TRACE
    if (p.trace && tile_id == p.n_blocks / 2 && threadIdx.x == 0) p.trace[0] = tr0;
    TILE_STAMP(1);                                  // loads issued
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TILE_STAMP(2);                                  // loads back
#endif
    // ---- level 0 -------------------------------------------------------------------------------------------------
    if constexpr (IN == T_PROLONG) {
        // U + P(coarse): doProlongation :700 as a gather, then doGridAddition :569 -- the streaming kernel's expressions:
        // hA / hB = (c1*(c2x-f_x) + c2*(f_x-c1x)) of the coarse rows owner / owner + 1 at this lane's column
        real_t hA = 0, hB = c0a * pc_hi + c0b * pc_lo;
        int c_row = own[0] - 1;
        const real_t c_dx = p.c_dx, c_rcp = p.c_dx_rcp;
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            if (own[j] != c_row) {  // wave-uniform: the owner row advanced by one
                hA = hB;
                c_row = own[j];
                hB = ca[j] * pc_hi + cb[j] * pc_lo;
            }
            const real_t own_yh = lane_value(t_rhi, j), own_yl = lane_value(t_rlo, j);
            const real_t num = hA * own_yh + hB * own_yl;
            const real_t pv = div_by_const(div_by_const(num, c_dx, c_rcp), c_dx, c_rcp);
            v[j] = v[j] + pv;
        }
    }

    // first / last row of the neighbouring waves' blocks, of the level in v[]
    int xb = 0;
    auto exchange = [&](const real_t &first, const real_t &last, real_t &below, real_t &above) {
        xch[xb][wave][0][lane] = first;
        xch[xb][wave][1][lane] = last;
        __syncthreads();
        below = wave > 0 ? xch[xb][wave - 1][1][lane] : real_t(0.0);
        above = wave < WAVES - 1 ? xch[xb][wave + 1][0][lane] : real_t(0.0);
        xb ^= 1;
    };

    // ---- S sweeps (src/MG_solver_CPU.cpp:587-599), all rows of the window at once -----------------------------------
#pragma unroll
    for (int s = 1; s <= S; ++s) {
        if (IN == T_ZERO && s == 1) {
            // the first sweep from the zero field: every neighbour and the point itself are +0, so the sum, `- 4*U` and
            // `U +` of the general expression are exact no-ops -- the same bits without them
#pragma unroll
            for (int j = 0; j < RPW; ++j) {
                const real_t t4 = real_t(0.0) - dx2 * f[j];
                v[j] = fused_mul_add(hi_bits_and(qc, row_inner(j)), t4, real_t(0.0));
            }
            continue;
        }
        real_t below, above;
        exchange(v[0], v[RPW - 1], below, above);
        real_t o[RPW];
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const real_t c = v[j];
            const real_t so = j > 0 ? v[j > 0 ? j - 1 : 0] : below;
            const real_t nw = j < RPW - 1 ? v[j < RPW - 1 ? j + 1 : 0] : above;
            const real_t w = from_lane_below(c), e = from_lane_above(c);
            // :590  U += 0.25*(U[i+1]+U[i-1]+U[j+1]+U[j-1] - 4U - dx^2 F), fused exactly as in the streaming kernel
            const real_t t4 = minus4(nw + so + e + w, c) - dx2 * f[j];
            o[j] = fused_mul_add(hi_bits_and(qc, row_inner(j)), t4, c);
        }
#pragma unroll
        for (int j = 0; j < RPW; ++j) v[j] = o[j];
    }

    TILE_STAMP(3);                                  // sweeps done
    // ---- the smoothed U ---------------------------------------------------------------------------------------------
    if (!p.no_out && lane_owns) {
#pragma unroll
        for (int j = 0; j < RPW; ++j)
            if (row_owned(j)) a_out[(size_t)(row_y(j) - p.row_base) * N + x] = v[j];
    }

    // ---- residual (:560), error sums (:610/:617), restriction (:656-678) -------------------------------------------
    if (want_res) {
        real_t below, above;
        exchange(v[0], v[RPW - 1], below, above);
        real_t d[RPW];
        double acc = 0.0;
        const real_t ms = col_edge ? real_t(0.0) : (p.d_sign < 0 ? real_t(-1.0) : real_t(1.0));
        // error norm: interior points with (row + col) even, each counted by the lane and tile that own it
        const bool mine = lane_owns && !col_edge;
        const int nm_even = (mine && ((x + yb) & 1) == 0) ? -1 : 0, nm_odd = (mine && ((x + yb) & 1) != 0) ? -1 : 0;
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const real_t c = v[j];
            const real_t so = j > 0 ? v[j > 0 ? j - 1 : 0] : below;
            const real_t nw = j < RPW - 1 ? v[j < RPW - 1 ? j + 1 : 0] : above;
            const real_t w = from_lane_below(c), e = from_lane_above(c);
            const real_t r = inv * minus4(nw + so + e + w, c) - f[j];
            const int inner = row_inner(j);
            if constexpr (RESTRICT) d[j] = r * hi_bits_and(ms, inner);  // sign flip :277-280 and the zero rim in one exact product
            const int am = ((j & 1) ? nm_odd : nm_even) & ((row_owned(j) && (unsigned)(yb + j - p.norm_y0) < (unsigned)(p.norm_y1 - p.norm_y0)) ? inner : 0);
            acc += fabs(bits_and((double)r, am));
        }
        if (a_part) {
            const double total = wave_sum_dpp(acc);
            if (lane == 0) a_part[(size_t)tile_id * WAVES + wave] = total;
        }
        if constexpr (RESTRICT) {
            // the row after the block's last one: the next wave's first
            real_t d_next, unused;
            exchange(d[0], d[RPW - 1], unused, d_next);
            (void)unused;
            // (wave-uniform tile conditions first: interior tiles skip the rim stores without touching the exec mask)
            const bool first_col_lane = tile_x == 0 && lane_owns && x == 0, last_col_lane = ox1 == N && lane_owns && x == N - 1;
            // the rim of the next level's F is zero (doRestriction's memset, :651): rim rows by the tiles that hold fine
            // rows 0 and N-1, rim columns by the lanes that own fine columns 0 and N-1 alongside every coarse row
            if (wave == 0) {
                for (int edge = 0; edge < 2; ++edge) {
                    if (edge == 0 ? (oy0 != 0) : (oy1 != N)) continue;
                    real_t *row = a_Fc + (size_t)((edge == 0 ? 0 : p.M - 1) - p.fc_base) * p.M;
                    if (rc_col >= 0) row[rc_col] = 0.0;
                    if (first_col_lane) row[0] = 0.0;
                    if (last_col_lane) row[p.M - 1] = 0.0;
                }
            }
#pragma unroll
            for (int j = 0; j < RPW; ++j) {
                const int rc_row = lane_value(t_rc, j);   // coarse row whose lower-left sample lies in fine row j of the block, or -1
                if (rc_row < 0) continue;
                const real_t wc = lane_value(t_rw, j), wd = real_t(1.0) - wc;  // c, d of :664-666
                const real_t u0 = d[j], u2 = j < RPW - 1 ? d[j < RPW - 1 ? j + 1 : 0] : d_next;
                const real_t u1 = from_lane_above(u0), u3 = from_lane_above(u2);
                // :676  U_c = b*d*U_f[f] + a*d*U_f[f+1] + c*b*U_f[f+N] + a*c*U_f[f+N+1]
                const real_t vc = rw_b * wd * u0 + rw_a * wd * u1 + wc * rw_b * u2 + rw_a * wc * u3;
                real_t *crow = a_Fc + (size_t)(rc_row - p.fc_base) * p.M;
                if (rc_col >= 0) crow[rc_col] = vc;
                if (tile_x == 0 && first_col_lane) crow[0] = 0.0;
                if (ox1 == N && last_col_lane) crow[p.M - 1] = 0.0;
            }
        }
    }
    TILE_STAMP(4);                                  // residual / restriction issued
#ifdef MG_TILE_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TILE_STAMP(5);                                  // stores acknowledged
#endif
}

// geometry of one instantiation, for the launcher
template <int S, int IN, bool RESTRICT, int RPW, int WAVES>
void launch_tile(hipStream_t s, TileParams p, double *err_out, bool raw_norm)
{
    constexpr int HALO = Geom<S, IN, RESTRICT>::HALO, TY = RPW * WAVES - 2 * HALO, TX = Geom<S, IN, RESTRICT>::TX;
    const int N = p.N;
    const int own = p.own_y1 - p.own_y0;
    if (own <= 0) return;
    p.tiles_x = (N + TX - 1) / TX;
    const int tiles_y = (own + TY - 1) / TY;
    p.n_blocks = p.tiles_x * tiles_y;
    p.part = nullptr;
    const size_t n_part = (size_t)p.n_blocks * WAVES;
    const int nb = p.batch ? p.n_batch : 1;
    if (err_out || (p.batch && p.err_outs)) {
        p.part = norm_partials(n_part * (size_t)nb);  // every wave of every tile writes its slot
        if (!p.part) return;
    }
    p.part_stride = (int)n_part;
    const int grid = ((p.n_blocks + 7) / 8) * 8;
#ifdef MG_TILE_TRACE
    static long long *trace_dev = nullptr;
    if (!trace_dev) (void)hipMalloc((void **)&trace_dev, 8 * sizeof(long long));
    (void)hipMemsetAsync(trace_dev, 0, 8 * sizeof(long long), s);
    p.trace = trace_dev;
#endif
    hipLaunchKernelGGL((k_jacobi_tile<S, IN, RESTRICT, RPW, WAVES>), dim3(grid, nb), dim3(64 * WAVES), 0, s, p);
#ifdef MG_TILE_TRACE
    {
        long long t[8];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(t, trace_dev, sizeof t, hipMemcpyDeviceToHost);
        fprintf(stderr, "[tile trace] N=%d S=%d IN=%d R=%d rpw=%d tiles=%d: to loads issued %.2f, loads back %.2f, sweeps %.2f, residual+restrict %.2f, stores %.2f us\n",
                N, S, IN, (int)RESTRICT, RPW, p.n_blocks, (t[1] - t[0]) * 0.01, (t[2] - t[1]) * 0.01, (t[3] - t[2]) * 0.01, (t[4] - t[3]) * 0.01,
                (t[5] - t[4]) * 0.01);
    }
#endif
    // a slab launch leaves its RAW partial sum; the caller combines the slabs in rank order
    if (p.batch && p.err_outs) {
        for (int i = 0; i < nb; ++i)
            if (p.err_outs[i]) norm_finish(s, p.part + (size_t)i * n_part, n_part, raw_norm ? -1 : N, p.err_outs[i]);
    } else if (err_out) {
        norm_finish(s, p.part, n_part, raw_norm ? -1 : N, err_out);
    }
}

// Window geometry: RPW rows per wave x WAVES waves.  The launch is bound by the instruction stream of a lone wave
// (~170 instructions per row of the window for three sweeps, residual and restriction; one wave per SIMD at these grid
// sizes), so the same window cut into MORE waves of FEWER rows is faster as long as the boundary-row exchange (two LDS
// writes, two reads and a barrier per sweep, whatever RPW) stays small against the rows' own work:
//   3 rows x 8 waves (24-row window) up to N = 512, 6 rows x 8 waves (48-row window) above.
// MG_TILE_GEOM=<rpw>x<waves> forces one of 3x8, 6x4, 6x8, 12x4 (A/B switch).
template <int S, int IN, bool RESTRICT>
void launch_geom(hipStream_t s, const TileParams &p, double *err_out, bool raw_norm)
{
    static const int forced = [] {
        const char *e = getenv("MG_TILE_GEOM");
        if (!e) return 0;
        int r = 0, w = 0;
        return sscanf(e, "%dx%d", &r, &w) == 2 ? r * 100 + w : 0;
    }();
    const int geom = forced ? forced : (p.N <= 512 ? 308 : 608);
    switch (geom) {
        case 604: launch_tile<S, IN, RESTRICT, 6, 4>(s, p, err_out, raw_norm); break;
        case 1204: launch_tile<S, IN, RESTRICT, 12, 4>(s, p, err_out, raw_norm); break;
        case 608: launch_tile<S, IN, RESTRICT, 6, 8>(s, p, err_out, raw_norm); break;
        default: launch_tile<S, IN, RESTRICT, 3, 8>(s, p, err_out, raw_norm); break;
    }
}

template <int S>
void launch_steps(hipStream_t s, const TileParams &p, double *err_out, bool raw_norm)
{
    const bool restrict_out = p.Fc != nullptr, prolong_in = p.coarse != nullptr, zero = p.in == nullptr;
    if (restrict_out) {
        if (zero) launch_geom<S, T_ZERO, true>(s, p, err_out, raw_norm);
        else launch_geom<S, T_LOAD, true>(s, p, err_out, raw_norm);
    } else if (prolong_in) {
        launch_geom<S, T_PROLONG, false>(s, p, err_out, raw_norm);
    } else {
        if (zero) launch_geom<S, T_ZERO, false>(s, p, err_out, raw_norm);
        else launch_geom<S, T_LOAD, false>(s, p, err_out, raw_norm);
    }
}

struct Tables {
    const int *p_orow = nullptr, *p_ocol = nullptr;
    const real_t *p_rhi = nullptr, *p_rlo = nullptr, *p_chi = nullptr, *p_clo = nullptr;
    real_t c_dx = 0, c_dx_rcp = 0;
    bool p_closed = false;   // ProlongTable::closed_form
    const int *r_inv = nullptr;
    const real_t *r_w = nullptr, *r_wf = nullptr;
};

constexpr int MAX_S = 4;

// one fused node on the whole grid: same contract as the streaming kernel's entry point without row windows, without a
// stored residual and without the recomputing form (the caller routes those to the streaming kernel)
inline void run(hipStream_t s, int N, real_t dx2, real_t inv, const real_t *in, const real_t *F, real_t *out, int steps, double *err_out,
                int d_sign, const real_t *coarse, int Nc, real_t *Fc, int M, const Tables &tb, bool no_out,
                const RowWindow *fine_w = nullptr, const RowWindow *coarse_w = nullptr, const RowWindow *fc_w = nullptr,
                const NodeBatch *batch = nullptr)
{
    if (batch && (fine_w || coarse_w || fc_w || batch->n < 1)) {
        fail(MG_ERR_ARG, "jacobi_tile: a batch of instances runs whole grids");
        return;
    }
    if (steps < 1 || steps > MAX_S || N < 8) {
        fail(MG_ERR_ARG, "jacobi_tile: %d sweeps on N=%d (1..%d sweeps, N >= 8)", steps, N, MAX_S);
        return;
    }
    if (coarse && Fc) {
        fail(MG_ERR_ARG, "jacobi_tile: a node either prolongs or restricts");
        return;
    }
    TileParams p = {};
    if (batch) {
        p.batch = batch->dev;
        p.n_batch = batch->n;
        p.err_outs = batch->err_outs;
    }
    p.N = N;
    p.dx2 = dx2;
    p.inv = inv;
    p.in = in;
    p.F = F;
    p.out = out;
    p.no_out = no_out ? 1 : 0;
    p.d_sign = d_sign;
    p.row_base = fine_w ? fine_w->base : 0;
    p.rows_local = fine_w ? fine_w->rows : N;
    p.own_y0 = fine_w ? fine_w->own_lo : 0;
    p.own_y1 = fine_w ? fine_w->own_hi : N;
    p.norm_y0 = fine_w && fine_w->norm_lo >= 0 ? fine_w->norm_lo : p.own_y0;
    p.norm_y1 = fine_w && fine_w->norm_lo >= 0 ? fine_w->norm_hi : p.own_y1;
    p.coarse_rows = 1;
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
        static const bool no_closed = getenv("MG_TILE_NO_CLOSED_FORM") != nullptr;   // A/B switch
        p.own_closed = (tb.p_closed && !no_closed && N <= 4096) ? 1 : 0;
        p.own_rcp = 1.0f / (float)(N - 1);
    }
    if (Fc) {
        p.fc_base = fc_w ? fc_w->base : 0;
        p.Fc = Fc;
        p.M = M;
        p.r_inv = tb.r_inv;
        p.r_w = tb.r_w;
        p.r_wf = tb.r_wf;
    }
    const bool raw_norm = fine_w != nullptr;
    switch (steps) {
        case 1: launch_steps<1>(s, p, err_out, raw_norm); break;
        case 2: launch_steps<2>(s, p, err_out, raw_norm); break;
        case 3: launch_steps<3>(s, p, err_out, raw_norm); break;
        default: launch_steps<4>(s, p, err_out, raw_norm); break;
    }
}

}  // namespace tile
}  // namespace MG_REAL_NS
}  // namespace k
}  // namespace mg
