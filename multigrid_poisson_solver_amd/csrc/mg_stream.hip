// mg_stream.hip -- temporally blocked, wave-streaming Jacobi smoother for CDNA4.
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
//   * the next PF rows are always in flight (register FIFO, loop unrolled by PF) so a
//     wave keeps 2*PF KiB of loads outstanding without relying on occupancy.
//
// Every point is updated with exactly the reference's expression and association order
// (-ffp-contract=off), so the result is bit-identical to S separate sweeps; a halo point
// computed twice gets the same bits twice.  Algorithmic traffic of S sweeps + residual:
// read U, read F, write U, write D = 32 B per point instead of 24*S + 24 + 8.
#include <hip/hip_runtime.h>

#include "mg_internal.h"

namespace mg {
namespace k {

namespace {

constexpr int PF = 4;            // rows of U and F in flight per lane
constexpr int WAVES_PER_WG = 4;  // 4 adjacent strips of one chunk
constexpr int MAX_S = 4;

typedef double double2_t __attribute__((ext_vector_type(2)));

struct StreamParams {
    int N;
    double dx2, inv;
    const double *in;   // nullptr: level 0 is all zero
    const double *F;
    double *out;
    double *D;          // nullptr: no residual output
    int d_sign;
    double *part;       // nullptr: no error norm; else one partial per wave
    int rows_per_chunk;
    int groups;         // workgroups per chunk row
    int n_blocks;       // chunks * groups
};

// value of the neighbouring lane (lane-1 / lane+1); lanes at the wave edge read 0
__device__ __forceinline__ double from_lane_below(double v)
{
    union { double d; int i[2]; } a, r;
    a.d = v;
    r.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    r.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], 0x138, 0xf, 0xf, false);
    return r.d;
}
__device__ __forceinline__ double from_lane_above(double v)
{
    union { double d; int i[2]; } a, r;
    a.d = v;
    r.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
    r.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], 0x130, 0xf, 0xf, false);
    return r.d;
}

template <int COLS>
struct Row {
    double v[COLS];
};

template <int COLS>
__device__ __forceinline__ Row<COLS> load_row(const double *__restrict__ base, bool ok)
{
    Row<COLS> r;
#pragma unroll
    for (int j = 0; j < COLS; ++j) r.v[j] = 0.0;
    if (ok) {
        if constexpr (COLS == 2) {
            const double2_t t = *reinterpret_cast<const double2_t *>(base);
            r.v[0] = t.x;
            r.v[1] = t.y;
        } else {
            r.v[0] = *base;
        }
    }
    return r;
}

template <int COLS>
__device__ __forceinline__ void store_row(double *__restrict__ base, const Row<COLS> &r)
{
    if constexpr (COLS == 2) {
        double2_t t;
        t.x = r.v[0];
        t.y = r.v[1];
        *reinterpret_cast<double2_t *>(base) = t;
    } else {
        *base = r.v[0];
    }
}

template <int S>
struct Halo {
    static constexpr int value = (S + 2) & ~1;  // >= S+1, even (16 B aligned strips)
};

template <int S, int COLS, bool ZERO_IN>
__global__ __launch_bounds__(64 * WAVES_PER_WG) void k_jacobi_stream(const StreamParams p)
{
    constexpr int W = 64 * COLS;
    constexpr int H = Halo<S>::value;
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
    if (own_x0 >= N) return;  // no barriers in this kernel: a wave may leave at any time
    const int y0 = chunk * p.rows_per_chunk;
    int y1 = y0 + p.rows_per_chunk;
    if (y1 > N) y1 = N;
    if (y0 >= N) return;

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

    const double dx2 = p.dx2, inv = p.inv;
    const bool want_res = p.D != nullptr || p.part != nullptr;

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

    const int y_first = y0 - (S + 1);           // first input row
    const int T = (y1 - y0) + 2 * (S + 1);      // input rows consumed
    const int y_end = y_first + T;              // one past the last input row
    const size_t col_off = (size_t)(xl < 0 ? 0 : xl);

    Row<COLS> pu[PF], pf[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) {
        const int y = y_first + k;
        const bool ok = lane_loads && y >= 0 && y < N && y < y_end;
        const size_t off = (size_t)(y < 0 ? 0 : y) * N + col_off;
        if constexpr (!ZERO_IN) pu[k] = load_row<COLS>(p.in + off, ok);
        pf[k] = load_row<COLS>(p.F + off, ok);
    }

    double acc = 0.0;

    for (int t0 = 0; t0 < T; t0 += PF) {
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int yin = y_first + t0 + k;
            Row<COLS> nw, cf = pf[k];
            if constexpr (ZERO_IN) {
#pragma unroll
                for (int j = 0; j < COLS; ++j) nw.v[j] = 0.0;
            } else {
                nw = pu[k];
            }
            {   // refill this FIFO slot with the row PF ahead
                const int y = yin + PF;
                const bool ok = lane_loads && y >= 0 && y < N && y < y_end;
                const size_t off = (size_t)(y < 0 ? 0 : y) * N + col_off;
                if constexpr (!ZERO_IN) pu[k] = load_row<COLS>(p.in + off, ok);
                pf[k] = load_row<COLS>(p.F + off, ok);
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
                const double west0 = from_lane_below(c.v[COLS - 1]);
                const double east_last = from_lane_above(c.v[0]);
                Row<COLS> o;
#pragma unroll
                for (int j = 0; j < COLS; ++j) {
                    const double w = j == 0 ? west0 : c.v[j > 0 ? j - 1 : 0];
                    const double e = j == COLS - 1 ? east_last : c.v[j < COLS - 1 ? j + 1 : 0];
                    // src/MG_solver_CPU.cpp:590: U += 0.25*(U[i+1]+U[i-1]+U[j+1]+U[j-1] - 4U - dx^2 F)
                    const double t = nw.v[j] + so.v[j] + e + w - 4 * c.v[j] - dx2 * fq[l].v[j];
                    const double u = c.v[j] + 0.25 * t;
                    o.v[j] = (row_edge || col_edge[j]) ? c.v[j] : u;
                }
                older[l - 1] = c;
                newer[l - 1] = nw;
                nw = o;
            }

            // nw is level S of row yin-S: the smoothed U
            {
                const int y = yin - S;
                if (y >= y0 && y < y1 && lane_owns) store_row<COLS>(p.out + (size_t)y * N + xl, nw);
            }

            // residual stage, row yin-S-1 (src/MG_solver_CPU.cpp:560 and the error sums :611)
            if (want_res) {
                const int y = yin - S - 1;
                const bool mine = y >= y0 && y < y1 && lane_owns;  // each point counted once
                const bool row_edge = y <= 0 || y >= N - 1;
                const Row<COLS> c = newer[S], so = older[S];
                const double west0 = from_lane_below(c.v[COLS - 1]);
                const double east_last = from_lane_above(c.v[0]);
                Row<COLS> d;
#pragma unroll
                for (int j = 0; j < COLS; ++j) {
                    const double w = j == 0 ? west0 : c.v[j > 0 ? j - 1 : 0];
                    const double e = j == COLS - 1 ? east_last : c.v[j < COLS - 1 ? j + 1 : 0];
                    const double r = inv * (nw.v[j] + so.v[j] + e + w - 4 * c.v[j]) - fq[S + 1].v[j];
                    const bool interior = !(row_edge || col_edge[j]);
                    const double dv = interior ? r : 0.0;
                    d.v[j] = p.d_sign < 0 ? -dv : dv;
                    // (row+col) even interior points only, :610/:617
                    if (mine && interior && (((y & 1) == 0) == col_even[j])) acc += fabs(r);
                }
                if (mine && p.D) store_row<COLS>(p.D + (size_t)y * N + xl, d);
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
// this instantiation x CUs) where the grid is large enough, never fewer than 32 rows per
// chunk (each chunk re-reads 2(S+1) halo rows), then the fixed-order error reduction.
template <int S, int COLS, bool ZERO_IN>
void launch_k(hipStream_t s, StreamParams p, double *err_out)
{
    static int blocks_per_cu = 0;
    if (blocks_per_cu == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_jacobi_stream<S, COLS, ZERO_IN>, 64 * WAVES_PER_WG, 0) != hipSuccess || n < 1) {
            (void)hipGetLastError();
            n = 2;
        }
        blocks_per_cu = n > 8 ? 8 : n;
    }
    const int N = p.N;
    constexpr int OW = 64 * COLS - 2 * Halo<S>::value;
    const int strips = (N + OW - 1) / OW;
    const int groups = (strips + WAVES_PER_WG - 1) / WAVES_PER_WG;
    const int resident = ctx().n_cu * blocks_per_cu;
    int chunks = resident / groups;
    const int max_chunks = (N + 31) / 32;
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks < 1) chunks = 1;
    const int rows = (N + chunks - 1) / chunks;
    chunks = (N + rows - 1) / rows;
    p.rows_per_chunk = rows;
    p.groups = groups;
    p.n_blocks = chunks * groups;
    p.part = nullptr;
    const size_t n_part = (size_t)p.n_blocks * WAVES_PER_WG;
    if (err_out) {
        p.part = partials(n_part);
        if (!p.part) return;
        // waves whose strip lies outside the grid exit without writing their slot
        (void)hipMemsetAsync(p.part, 0, n_part * sizeof(double), s);
    }
    const int grid = ((p.n_blocks + 7) / 8) * 8;
    hipLaunchKernelGGL((k_jacobi_stream<S, COLS, ZERO_IN>), dim3(grid), dim3(64 * WAVES_PER_WG), 0, s, p);
    if (err_out) finish_smoothing_error(s, p.part, n_part, N, err_out);
}

template <int COLS>
void launch(hipStream_t s, const StreamParams &p, int steps, double *err_out)
{
    const bool z = p.in == nullptr;
    switch (steps) {
        case 1: z ? launch_k<1, COLS, true>(s, p, err_out) : launch_k<1, COLS, false>(s, p, err_out); break;
        case 2: z ? launch_k<2, COLS, true>(s, p, err_out) : launch_k<2, COLS, false>(s, p, err_out); break;
        case 3: z ? launch_k<3, COLS, true>(s, p, err_out) : launch_k<3, COLS, false>(s, p, err_out); break;
        default: z ? launch_k<4, COLS, true>(s, p, err_out) : launch_k<4, COLS, false>(s, p, err_out); break;
    }
}

}  // namespace

int stream_max_steps() { return MAX_S; }
bool stream_supported(int N) { return N >= 3; }

void jacobi_stream(hipStream_t s, int N, double dx2, double inv, const double *in, const double *F, double *out,
                   int steps, double *err_out, double *D_out, int d_sign, const double *, int, const ProlongTable *)
{
    if (steps < 1 || steps > MAX_S) {
        fail(MG_ERR_ARG, "jacobi_stream: %d sweeps per launch (1..%d)", steps, MAX_S);
        return;
    }
    StreamParams p;
    p.N = N;
    p.dx2 = dx2;
    p.inv = inv;
    p.in = in;
    p.F = F;
    p.out = out;
    p.D = D_out;
    p.d_sign = d_sign;
    p.part = nullptr;
    p.rows_per_chunk = p.groups = p.n_blocks = 0;
    if (N % 2 == 0) launch<2>(s, p, steps, err_out);  // 16 B lanes need an even row pitch
    else launch<1>(s, p, steps, err_out);
}

}  // namespace k
}  // namespace mg
