// mg_kernels.hip -- fp64 HIP kernels (gfx950 / CDNA4, wave64) for the multigrid
// operators, plus their launchers.  Built with -ffp-contract=off: every expression
// keeps the association order of the reference CPU code so the output arrays are
// bit-identical (hard part 4).
//
// Layout: row-major N x N doubles, index = col + N*row, boundary included
// (src/MG_solver_CPU.cpp:484).  All kernels are HBM-bound streaming kernels: lanes map
// to consecutive columns so every wave instruction touches one contiguous 512 B
// segment of a row.  The temporally blocked smoother lives in mg_stream.hip.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "mg_divconst.h"
#include "mg_gs_wave.h"
#include "mg_exp_table.h"
#include "mg_internal.h"

namespace mg {
namespace k {

namespace {

constexpr int TB = 256;       // threads per block for the streaming kernels
constexpr int ROWS_PB = 4;    // rows per block

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// sum over the block in a fixed order; valid in thread 0
__device__ __forceinline__ double block_sum(double v)
{
    __shared__ double sm[16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v = wave_sum(v);
    __syncthreads();  // protect sm against a previous use
    if (lane == 0) sm[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; ++i) r += sm[i];
    }
    return r;
}

__device__ __forceinline__ bool rim(int r, int c, int N)
{
    return r == 0 || c == 0 || r == N - 1 || c == N - 1;
}

// 5-point sum in the reference's order: row+1, row-1, col+1, col-1, then -4*centre
// (src/MG_solver_CPU.cpp:560,:590,:611)
__device__ __forceinline__ double star_minus4(const double *__restrict__ A, size_t p, int N)
{
    return A[p + N] + A[p - N] + A[p + 1] + A[p - 1] - 4 * A[p];
}

// ---------------------------------------------------------------- Jacobi, one sweep
// src/MG_solver_CPU.cpp:587-599: U = U_old + 0.25*(star(U_old) - 4 U_old - dx^2 F)
template <bool ZERO_IN>
__global__ __launch_bounds__(TB) void k_jacobi_simple(int N, double dx2, const double *__restrict__ in,
                                                      const double *__restrict__ F, double *__restrict__ out)
{
    const int c = blockIdx.x * TB + threadIdx.x;
    if (c >= N) return;
    const int r0 = blockIdx.y * ROWS_PB;
#pragma unroll
    for (int k = 0; k < ROWS_PB; ++k) {
        const int r = r0 + k;
        if (r >= N) return;
        const size_t p = (size_t)r * N + c;
        double v;
        if (ZERO_IN) {
            v = rim(r, c, N) ? 0.0 : 0.0 + 0.25 * (0.0 - dx2 * F[p]);
        } else {
            v = in[p];
            if (!rim(r, c, N)) v = v + 0.25 * (star_minus4(in, p, N) - dx2 * F[p]);
        }
        out[p] = v;
    }
}

// the same sweep for even N with 16 B per lane: a thread owns two adjacent points, reads the
// rows above/below as aligned pairs and its two outer neighbours as single doubles (L1 hits of
// the neighbouring lanes' pairs); one row per block, a massive grid of short blocks -- the
// shape that streams closest to the HBM ceiling on MI355X (scripts/ubench/stream_ceiling.hip)
typedef double double2_k __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(TB) void k_jacobi_pair(int N, double dx2, const double *__restrict__ in,
                                                    const double *__restrict__ F, double *__restrict__ out)
{
    const int c = 2 * (blockIdx.x * TB + threadIdx.x);
    const int r = blockIdx.y;
    if (c >= N) return;
    const size_t p = (size_t)r * N + c;
    const double2_k ctr = *reinterpret_cast<const double2_k *>(in + p);
    double2_k o = ctr;
    if (r > 0 && r < N - 1) {
        const double2_k up = *reinterpret_cast<const double2_k *>(in + p + N);
        const double2_k dn = *reinterpret_cast<const double2_k *>(in + p - N);
        const double2_k f = *reinterpret_cast<const double2_k *>(F + p);
        if (c > 0) {  // point c: west neighbour is the previous lane's second value
            const double w = in[p - 1];
            o.x = ctr.x + 0.25 * (up.x + dn.x + ctr.y + w - 4 * ctr.x - dx2 * f.x);
        }
        if (c + 1 < N - 1) {  // point c+1: east neighbour is the next lane's first value
            const double e = in[p + 2];
            o.y = ctr.y + 0.25 * (up.y + dn.y + e + ctr.x - 4 * ctr.y - dx2 * f.y);
        }
    }
    __builtin_nontemporal_store(o, reinterpret_cast<double2_k *>(out + p));
}

// The same with PR rows per thread and a rolling window of three row pairs in registers: every row of `in` is read once as
// the centre of its own row and stays in registers as the neighbour of the rows above and below (the one-row-per-block form
// reads it three times, twice through L2), F is read once with a non-temporal load.  Same expressions, same bits.
#ifndef MG_PAIR_ROWS
#define MG_PAIR_ROWS 4
#endif
constexpr int PR = MG_PAIR_ROWS;
__global__ __launch_bounds__(TB) void k_jacobi_pair_rows(int N, double dx2, const double *__restrict__ in,
                                                         const double *__restrict__ F, double *__restrict__ out)
{
    const int c = 2 * (blockIdx.x * TB + threadIdx.x);
    const int r0 = blockIdx.y * PR;
    if (c >= N) return;
    auto row_pair = [&](int r) {
        r = r < 0 ? 0 : (r < N ? r : N - 1);   // (rows beyond the grid: clamped, never used -- rows 0 and N-1 keep their value)
        return *reinterpret_cast<const double2_k *>(in + (size_t)r * N + c);
    };
    double2_k dn = row_pair(r0 - 1), ctr = row_pair(r0);
#pragma unroll
    for (int k = 0; k < PR; ++k) {
        const int r = r0 + k;
        if (r >= N) break;
        const double2_k up = row_pair(r + 1);
        const size_t p = (size_t)r * N + c;
        double2_k o = ctr;
        if (r > 0 && r < N - 1) {
            const double2_k f = __builtin_nontemporal_load(reinterpret_cast<const double2_k *>(F + p));
            if (c > 0) {
                const double w = in[p - 1];
                o.x = ctr.x + 0.25 * (up.x + dn.x + ctr.y + w - 4 * ctr.x - dx2 * f.x);
            }
            if (c + 1 < N - 1) {
                const double e = in[p + 2];
                o.y = ctr.y + 0.25 * (up.y + dn.y + e + ctr.x - 4 * ctr.y - dx2 * f.y);
            }
        }
        __builtin_nontemporal_store(o, reinterpret_cast<double2_k *>(out + p));
        dn = ctr;
        ctr = up;
    }
}

// ---------------------------------------------------------------- residual
// src/MG_solver_CPU.cpp:554-564 (and the driver's sign flip :277-280 when sign < 0)
__global__ __launch_bounds__(TB) void k_residual(int N, double inv, const double *__restrict__ U,
                                                 const double *__restrict__ F, double *__restrict__ D, int sign)
{
    const int c = blockIdx.x * TB + threadIdx.x;
    if (c >= N) return;
    const int r0 = blockIdx.y * ROWS_PB;
#pragma unroll
    for (int k = 0; k < ROWS_PB; ++k) {
        const int r = r0 + k;
        if (r >= N) return;
        const size_t p = (size_t)r * N + c;
        double v = 0.0;
        if (!rim(r, c, N)) v = inv * star_minus4(U, p, N) - F[p];
        D[p] = sign < 0 ? -v : v;
    }
}

// getResidual on large even grids: 16 B per lane, PR rows per thread with a rolling window of three row pairs (every row of U
// read once), F and D through non-temporal accesses -- the shape of k_jacobi_pair_rows.  Same expression (star_minus4's order).
__global__ __launch_bounds__(TB) void k_residual_pairs(int N, double inv, const double *__restrict__ U,
                                                       const double *__restrict__ F, double *__restrict__ D, int sign)
{
    const int c = 2 * (blockIdx.x * TB + threadIdx.x);
    const int r0 = blockIdx.y * PR;
    if (c >= N) return;
    const int cl = c > 0 ? c - 1 : 0, cr = c + 2 < N ? c + 2 : N - 1;
    auto row_pair = [&](int r) {
        r = r < 0 ? 0 : (r < N ? r : N - 1);
        return *reinterpret_cast<const double2_k *>(U + (size_t)r * N + c);
    };
    double2_k up = row_pair(r0 - 1), mid = row_pair(r0);
#pragma unroll
    for (int k = 0; k < PR; ++k) {
        const int r = r0 + k;
        if (r >= N) break;
        const double2_k down = row_pair(r + 1);
        const size_t p = (size_t)r * N + c;
        double2_k v = {0.0, 0.0};
        if (r > 0 && r < N - 1) {
            const double left = U[(size_t)r * N + cl], right = U[(size_t)r * N + cr];
            const double2_k f = __builtin_nontemporal_load(reinterpret_cast<const double2_k *>(F + p));
            if (c > 0) v.x = inv * (down.x + up.x + mid.y + left - 4 * mid.x) - f.x;
            if (c + 1 < N - 1) v.y = inv * (down.y + up.y + right + mid.x - 4 * mid.y) - f.y;
        }
        if (sign < 0) v = -v;
        __builtin_nontemporal_store(v, reinterpret_cast<double2_k *>(D + p));
        up = mid;
        mid = down;
    }
}

// ---------------------------------------------------------------- smoothing error
// src/MG_solver_CPU.cpp:607-622: both sums run over (row+col) even interior points
__global__ __launch_bounds__(TB) void k_smoothing_error(int N, double inv, const double *__restrict__ U,
                                                        const double *__restrict__ F, double *__restrict__ part)
{
    const int c = blockIdx.x * TB + threadIdx.x;
    const int r0 = blockIdx.y * ROWS_PB;
    double acc = 0.0;
    if (c < N) {
#pragma unroll
        for (int k = 0; k < ROWS_PB; ++k) {
            const int r = r0 + k;
            if (r < N && !rim(r, c, N) && ((r + c) & 1) == 0) {
                const size_t p = (size_t)r * N + c;
                acc += fabs(inv * star_minus4(U, p, N) - F[p]);
            }
        }
    }
    const double s = block_sum(acc);
    if (threadIdx.x == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = s;
}

// mixed-precision refinement: the fp64 residual of the fp64 iterate, handed to the fp32 cycle as its
// source, src = (float)(-(A U - F)), and the doSmoothing error metric (:607-622) of the iterate
__global__ __launch_bounds__(TB) void k_refine_residual(int N, double inv, const double *__restrict__ U,
                                                        const double *__restrict__ F, float *__restrict__ src,
                                                        double *__restrict__ part)
{
    const int c = blockIdx.x * TB + threadIdx.x;
    const int r0 = blockIdx.y * ROWS_PB;
    double acc = 0.0;
    if (c < N) {
#pragma unroll
        for (int k = 0; k < ROWS_PB; ++k) {
            const int r = r0 + k;
            if (r >= N) break;
            const size_t p = (size_t)r * N + c;
            double v = 0.0;
            if (!rim(r, c, N)) {
                v = inv * star_minus4(U, p, N) - F[p];
                if (((r + c) & 1) == 0) acc += fabs(v);
            }
            src[p] = (float)(-v);
        }
    }
    const double s = block_sum(acc);
    if (threadIdx.x == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = s;
}

// The same for even N with 16 B per lane: a thread owns two adjacent columns and walks RR rows with a rolling window of
// three row pairs in registers (every row of U is read once as aligned pairs; the two outer neighbours are single doubles,
// L1 hits of the neighbouring lanes' pairs); F and the result through non-temporal accesses: at N = 32768 the arrays are
// 4-8 GiB each and every byte of them is touched once.  The expressions are k_refine_residual's (star_minus4's order: row+1, row-1, col+1, col-1).
constexpr int RR = 4;   // (as k_jacobi_pair_rows: 8 rows per thread lose)
typedef double dpair_t __attribute__((ext_vector_type(2)));
typedef float fpair_t __attribute__((ext_vector_type(2)));
// (base / own_lo / own_hi: the arrays' first row and the rows to form -- the whole grid, or the window of a row slab, whose
// halo rows own_lo - 1 and own_hi of U are there)
__global__ __launch_bounds__(TB) void k_refine_residual_pairs(int N, double inv, const double *__restrict__ U,
                                                              const double *__restrict__ F, float *__restrict__ src,
                                                              double *__restrict__ part, int base, int own_lo, int own_hi)
{
    const int c2 = blockIdx.x * TB + threadIdx.x;   // column pair
    const int c = 2 * c2;
    const int r0 = own_lo + blockIdx.y * RR;
    double acc = 0.0;
    if (c < N) {
        const int cl = c > 0 ? c - 1 : 0, cr = c + 2 < N ? c + 2 : N - 1;
        const int lo = own_lo > 0 ? own_lo - 1 : 0, hi = own_hi < N ? own_hi : N - 1;   // rows of U that exist for this launch
        auto row_pair = [&](int r) {
            r = r < lo ? lo : (r < hi ? r : hi);
            return *reinterpret_cast<const dpair_t *>(U + (size_t)(r - base) * N + c);   // (plain: the block's halo rows are its neighbours' rows)
        };
        dpair_t up = row_pair(r0 - 1), mid = row_pair(r0);
#pragma unroll
        for (int k = 0; k < RR; ++k) {
            const int r = r0 + k;
            if (r >= own_hi) break;
            const dpair_t down = row_pair(r + 1);
            const size_t p = (size_t)(r - base) * N + c;
            const double left = U[(size_t)(r - base) * N + cl], right = U[(size_t)(r - base) * N + cr];
            const dpair_t f = __builtin_nontemporal_load(reinterpret_cast<const dpair_t *>(F + p));
            double v0 = 0.0, v1 = 0.0;
            if (!rim(r, c, N)) {
                v0 = inv * (down.x + up.x + mid.y + left - 4 * mid.x) - f.x;
                if (((r + c) & 1) == 0) acc += fabs(v0);
            }
            if (!rim(r, c + 1, N)) {
                v1 = inv * (down.y + up.y + right + mid.x - 4 * mid.y) - f.y;
                if (((r + c + 1) & 1) == 0) acc += fabs(v1);
            }
            fpair_t o;
            o.x = (float)(-v0);
            o.y = (float)(-v1);
            __builtin_nontemporal_store(o, reinterpret_cast<fpair_t *>(src + p));
            up = mid;
            mid = down;
        }
    }
    const double s = block_sum(acc);
    if (threadIdx.x == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = s;
}

// the same on a row window (slab mode): rows [own_lo, own_hi) of arrays whose first row is `base`; the
// halo rows own_lo - 1 and own_hi of U must be there; the norm is left as this slab's raw sum
__global__ __launch_bounds__(TB) void k_refine_residual_rows(int N, double inv, const double *__restrict__ U,
                                                             const double *__restrict__ F, float *__restrict__ src, int base,
                                                             int own_lo, int own_hi, double *__restrict__ part)
{
    const int c = blockIdx.x * TB + threadIdx.x;
    const int r0 = own_lo + blockIdx.y * ROWS_PB;
    double acc = 0.0;
    if (c < N) {
#pragma unroll
        for (int k = 0; k < ROWS_PB; ++k) {
            const int r = r0 + k;
            if (r >= own_hi) break;
            const size_t p = (size_t)(r - base) * N + c;
            double v = 0.0;
            if (!rim(r, c, N)) {
                v = inv * star_minus4(U, p, N) - F[p];
                if (((r + c) & 1) == 0) acc += fabs(v);
            }
            src[p] = (float)(-v);
        }
    }
    const double s = block_sum(acc);
    if (threadIdx.x == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = s;
}

// U += (double)e: the fp64 correction step of the refinement
__global__ __launch_bounds__(TB) void k_add_widened(double *__restrict__ U, const float *__restrict__ e, size_t n)
{
    const size_t stride = (size_t)gridDim.x * TB;
    for (size_t i = (size_t)blockIdx.x * TB + threadIdx.x; i < n; i += stride) U[i] = U[i] + (double)e[i];
}

// ... two elements per lane (16 B of U), one shot, non-temporal: every byte of the 12 GiB is touched once
__global__ __launch_bounds__(TB) void k_add_widened_pairs(double *__restrict__ U, const float *__restrict__ e, size_t n2)
{
    const size_t i = (size_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n2) return;
    dpair_t u = __builtin_nontemporal_load(reinterpret_cast<const dpair_t *>(U) + i);
    const fpair_t d = __builtin_nontemporal_load(reinterpret_cast<const fpair_t *>(e) + i);
    u.x = u.x + (double)d.x;
    u.y = u.y + (double)d.y;
    __builtin_nontemporal_store(u, reinterpret_cast<dpair_t *>(U) + i);
}

enum FinishMode { FIN_SMOOTH_ERR = 0, FIN_MEAN_NN = 1, FIN_RAW = 2 };

// second stage of every norm: fixed-order sum of the per-block partials
__global__ __launch_bounds__(1024) void k_finish(const double *__restrict__ part, size_t n, int mode, int N,
                                                 double *__restrict__ out)
{
    double acc = 0.0;
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) acc += part[i];
    const double s = block_sum(acc);
    if (threadIdx.x == 0) {
        double e = s;
        if (mode == FIN_SMOOTH_ERR) {  // *error = sum1+sum2; *error = *error/N/N  (:621-622)
            e = s + s;
            e = e / N / N;
        } else if (mode == FIN_MEAN_NN) {  // MGerror / (double)(N*N)  (:445)
            e = s / (double)(N * N);
        }
        *out = e;
    }
}

// many doSmoothing error reductions in one launch: block b finishes descriptor b
// (16 waves and four independent partial sums per thread: the finest levels leave ~8000 partials each, and this launch
// closes every window -- 7.3 us with 256 threads and one dependent chain of loads per thread)
__global__ __launch_bounds__(1024) void k_finish_batch(const NormBatch b)
{
    const int d = blockIdx.x;
    const double *__restrict__ part = b.part[d];
    const int n = b.n[d], N = b.N[d];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int i = threadIdx.x;
    for (; i + 3 * 1024 < n; i += 4 * 1024) {
        a0 += part[i];
        a1 += part[i + 1024];
        a2 += part[i + 2 * 1024];
        a3 += part[i + 3 * 1024];
    }
    for (; i < n; i += 1024) a0 += part[i];
    const double acc = (a0 + a1) + (a2 + a3);
    const double s = block_sum(acc);
    if (threadIdx.x == 0) {
        double e = s;
        if (N > 0) {  // N <= 0: raw partial sum of one row slab, combined across ranks later
            e = s + s;  // *error = sum1+sum2; *error = *error/N/N  (:621-622)
            e = e / N / N;
        }
        *b.out[d] = e;
    }
}

// ---------------------------------------------------------------- restriction
// src/MG_solver_CPU.cpp:656-678 with the 1-D tables built on the host from the
// reference's floor/fmod expressions (:661-666); rim of the coarse grid is 0 (:651)
// T = field type (double; float for the operator-by-operator path of the mixed-precision mode)
template <typename T>
__global__ __launch_bounds__(TB) void k_restrict(int N, const T *__restrict__ Uf, int M, T *__restrict__ Uc,
                                                 const int *__restrict__ lo, const T *__restrict__ w, int sign)
{
    const int cc = blockIdx.x * TB + threadIdx.x;
    const int rc = blockIdx.y;
    if (cc >= M) return;
    T v = 0.0;
    if (!rim(rc, cc, M)) {
        const T a = w[cc], b = T(1.0) - a;
        const T c = w[rc], d = T(1.0) - c;
        const size_t f = (size_t)lo[cc] + (size_t)lo[rc] * N;
        v = b * d * Uf[f] + a * d * Uf[f + 1] + c * b * Uf[f + N] + a * c * Uf[f + N + 1];
        if (sign < 0) v = -v;
    }
    Uc[(size_t)rc * M + cc] = v;
}

// ---------------------------------------------------------------- prolongation
// src/MG_solver_CPU.cpp:688-700 turned into a gather over fine points: the owning
// coarse cell and the four 1-D weight factors come from host tables that replay the
// reference's ceil() ranges and last-row/column rules (:697-718).
// doProlongation on large even fine grids whose every point has an owner cell (ProlongTable::fusable): two adjacent fine
// columns (16 B) per lane, the result through a non-temporal store, `.../c_dx/c_dx` (:700) through the correctly rounded
// division by a constant of mg_divconst.h -- the same bits as the IEEE division of k_prolong below, which stays the form the
// small grids (and with them the operator tests against the oracle) run; the reference's own checksums pin this one at
// 8192 ... 32768 (tests/test_parity_gpu.py).
template <bool ADD>
__global__ __launch_bounds__(TB) void k_prolong_pairs(int N, const double *__restrict__ Uc, int M, const double *__restrict__ Uf_in,
                                                      double *__restrict__ Uf_out, const int *__restrict__ orow,
                                                      const int *__restrict__ ocol, const double *__restrict__ row_hi,
                                                      const double *__restrict__ row_lo, const double *__restrict__ col_hi,
                                                      const double *__restrict__ col_lo, double c_dx, double c_rcp)
{
    const int l = 2 * (blockIdx.x * TB + threadIdx.x);
    const int kf = blockIdx.y;
    if (l >= M) return;
    const int i = orow[kf];
    const double yh = row_hi[kf], yl = row_lo[kf];
    const size_t q = (size_t)kf * M + l;
    const int j0 = ocol[l], j1 = ocol[l + 1];
    const double2_k xh = *reinterpret_cast<const double2_k *>(col_hi + l), xl = *reinterpret_cast<const double2_k *>(col_lo + l);
    const double *cu = Uc + (size_t)i * N, *cd = cu + N;
    const double a1 = cu[j0], a2 = cu[j0 + 1], a3 = cd[j0], a4 = cd[j0 + 1];
    const double b1 = cu[j1], b2 = cu[j1 + 1], b3 = cd[j1], b4 = cd[j1 + 1];
    double2_k v;
    v.x = div_by_const(div_by_const((a1 * xh.x + a2 * xl.x) * yh + (a3 * xh.x + a4 * xl.x) * yl, c_dx, c_rcp), c_dx, c_rcp);
    v.y = div_by_const(div_by_const((b1 * xh.y + b2 * xl.y) * yh + (b3 * xh.y + b4 * xl.y) * yl, c_dx, c_rcp), c_dx, c_rcp);
    if (ADD) {   // doGridAddition :569: U1 = U1 + U2
        const double2_k u = __builtin_nontemporal_load(reinterpret_cast<const double2_k *>(Uf_in + q));
        v.x = u.x + v.x;
        v.y = u.y + v.y;
    }
    __builtin_nontemporal_store(v, reinterpret_cast<double2_k *>(Uf_out + q));
}

template <bool ADD, typename T>
__global__ __launch_bounds__(TB) void k_prolong(int N, const T *__restrict__ Uc, int M, const T *__restrict__ Uf_in,
                                                T *__restrict__ Uf_out, const int *__restrict__ orow,
                                                const int *__restrict__ ocol, const T *__restrict__ row_hi,
                                                const T *__restrict__ row_lo, const T *__restrict__ col_hi,
                                                const T *__restrict__ col_lo, T c_dx)
{
    const int l = blockIdx.x * TB + threadIdx.x;
    const int kf = blockIdx.y;
    if (l >= M) return;
    const int i = orow[kf], j = ocol[l];
    const size_t q = (size_t)kf * M + l;
    if (i < 0 || j < 0) {  // no coarse cell writes this point (never for M >= N)
        if (ADD) Uf_out[q] = Uf_in[q];
        return;
    }
    const size_t p = (size_t)i * N + j;
    const T c1 = Uc[p], c2 = Uc[p + 1], c3 = Uc[p + N], c4 = Uc[p + N + 1];
    const T xh = col_hi[l], xl = col_lo[l], yh = row_hi[kf], yl = row_lo[kf];
    const T v = ((c1 * xh + c2 * xl) * yh + (c3 * xh + c4 * xl) * yl) / c_dx / c_dx;
    Uf_out[q] = ADD ? Uf_in[q] + v : v;   // doGridAddition :569: U1 = U1 + U2
}

// ---------------------------------------------------------------- elementwise
__global__ __launch_bounds__(TB) void k_add(size_t n, double *__restrict__ a, const double *__restrict__ b)
{
    for (size_t i = (size_t)blockIdx.x * TB + threadIdx.x; i < n; i += (size_t)gridDim.x * TB) a[i] = a[i] + b[i];
}
__global__ __launch_bounds__(TB) void k_negate(size_t n, double *__restrict__ a)
{
    for (size_t i = (size_t)blockIdx.x * TB + threadIdx.x; i < n; i += (size_t)gridDim.x * TB) a[i] = -a[i];
}
// the same on arrays far larger than the caches: two elements (16 B) per lane, one shot, non-temporal accesses
__global__ __launch_bounds__(TB) void k_add_pairs(size_t n2, double *__restrict__ a, const double *__restrict__ b)
{
    const size_t i = (size_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n2) return;
    double2_k x = __builtin_nontemporal_load(reinterpret_cast<const double2_k *>(a) + i);
    const double2_k y = __builtin_nontemporal_load(reinterpret_cast<const double2_k *>(b) + i);
    x.x = x.x + y.x;
    x.y = x.y + y.y;
    __builtin_nontemporal_store(x, reinterpret_cast<double2_k *>(a) + i);
}
__global__ __launch_bounds__(TB) void k_negate_pairs(size_t n2, double *__restrict__ a)
{
    const size_t i = (size_t)blockIdx.x * TB + threadIdx.x;
    if (i >= n2) return;
    double2_k x = __builtin_nontemporal_load(reinterpret_cast<const double2_k *>(a) + i);
    x = -x;
    __builtin_nontemporal_store(x, reinterpret_cast<double2_k *>(a) + i);
}

// fp64 <-> fp32 (mixed-precision mode): round to nearest / exact widening
__global__ __launch_bounds__(TB) void k_to_f32(float *__restrict__ dst, const double *__restrict__ src, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * TB + threadIdx.x; i < n; i += (size_t)gridDim.x * TB) dst[i] = (float)src[i];
}
__global__ __launch_bounds__(TB) void k_to_f64(double *__restrict__ dst, const float *__restrict__ src, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * TB + threadIdx.x; i < n; i += (size_t)gridDim.x * TB) dst[i] = (double)src[i];
}

// ---------------------------------------------------------------- problem definition
// src/MG_solver_CPU.cpp:488 / :544 call libm's exp().  exp_libm() evaluates it with the operations of the
// algorithm glibc has used since 2.28 (exp(x) = 2^(k/128) * exp(r): a 128-entry table, a degree-5 polynomial),
// in the FMA form glibc selects on every CPU with FMA: the fused multiply-adds below are the ones that build
// performs (disassembly of libm.so.6 2.35: z + Shift, both reduction steps, every polynomial step and the final
// scale are fused).  The table comes from scripts/gen_exp_table.py (computed from its definition), the constants
// are the algorithm's published ones.  Nothing is assumed: mg_source_selfcheck() compares getSource on the device
// with the host's libm bit for bit before the device form becomes the default (mg_abi.cpp).
__device__ const uint64_t exp_tab[256] = {MG_EXP_TABLE_VALUES};
__device__ __forceinline__ double exp_libm(double x)
{
    const uint32_t abstop = (uint32_t)(__double_as_longlong(x) >> 52) & 0x7ff;
    if (abstop - 0x3c9u > 0x3eu) {
        if (abstop < 0x3c9u) return 1.0 + x;  // |x| < 2^-54
        return exp(x);                        // |x| >= 512: the special-case paths are not reproduced (never reached on the unit square)
    }
    const double InvLn2N = 0x1.71547652b82fep+7, Shift = 0x1.8p+52, NegLn2hiN = -0x1.62e42fefa0000p-8,
                 NegLn2loN = -0x1.cf79abc9e3b3ap-47, C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3,
                 C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;
    const double kd_s = __builtin_fma(x, InvLn2N, Shift);
    const uint64_t ki = (uint64_t)__double_as_longlong(kd_s);
    const double kd = kd_s - Shift;
    double r = __builtin_fma(kd, NegLn2hiN, x);
    r = __builtin_fma(kd, NegLn2loN, r);
    const unsigned idx = 2u * (unsigned)(ki & 127u);
    const double tail = __longlong_as_double((long long)exp_tab[idx]);
    const uint64_t sbits = exp_tab[idx + 1] + (ki << 45);
    const double p23 = __builtin_fma(r, C3, C2);
    const double rt = r + tail;
    const double r2 = r * r;
    const double p45 = __builtin_fma(r, C5, C4);
    const double t = __builtin_fma(p23, r2, rt);
    const double r4 = r2 * r2;
    const double tmp = __builtin_fma(r4, p45, t);
    const double scale = __longlong_as_double((long long)sbits);
    return __builtin_fma(scale, tmp, scale);
}
// F points at grid row row0 (row slabs evaluate their own window)
__global__ __launch_bounds__(TB) void k_source(int N, double h, double *__restrict__ F, double min_x, double min_y, int row0)
{
    const int c = blockIdx.x * TB + threadIdx.x, r = blockIdx.y + row0;
    if (c >= N) return;
    double v = 0.0;
    if (!rim(r, c, N)) {
        const double x = (double)c * h + min_x, y = (double)r * h + min_y;
        v = 2.0 * x * (y - 1) * (y - 2.0 * x + x * y + 2.0) * exp_libm(x - y);
    }
    F[(size_t)blockIdx.y * N + c] = v;
}
__device__ __forceinline__ double analytic_at(int r, int c, int N, double h, double min_x, double min_y)
{
    if (rim(r, c, N)) return 0.0;
    const double x = (double)c * h + min_x, y = (double)r * h + min_y;
    return exp_libm(x - y) * x * (1.0 - x) * y * (1.0 - y);
}
__global__ __launch_bounds__(TB) void k_analytic(int N, double h, double *__restrict__ U, double min_x, double min_y)
{
    const int c = blockIdx.x * TB + threadIdx.x, r = blockIdx.y;
    if (c >= N) return;
    U[(size_t)r * N + c] = analytic_at(r, c, N, h, min_x, min_y);
}
__global__ __launch_bounds__(TB) void k_analytic_error(int N, double h, const double *__restrict__ U, double min_x,
                                                       double min_y, double *__restrict__ part)
{
    const int c = blockIdx.x * TB + threadIdx.x;
    const int r0 = blockIdx.y * ROWS_PB;
    double acc = 0.0;
    if (c < N) {
#pragma unroll
        for (int k = 0; k < ROWS_PB; ++k) {
            const int r = r0 + k;
            if (r < N) acc += fabs(analytic_at(r, c, N, h, min_x, min_y) - U[(size_t)r * N + c]);
        }
    }
    const double s = block_sum(acc);
    if (threadIdx.x == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = s;
}

// sum |analytic - U| over the owned rows of a row window (src/MG_solver_CPU.cpp:441-444)
__global__ __launch_bounds__(TB) void k_analytic_error_rows(int N, double h, const double *__restrict__ U, int base,
                                                            int own_lo, double min_x, double min_y,
                                                            int own_hi, double *__restrict__ part)
{
    const int c = blockIdx.x * TB + threadIdx.x;
    const int r0 = own_lo + blockIdx.y * ROWS_PB;
    double acc = 0.0;
    if (c < N) {
#pragma unroll
        for (int k = 0; k < ROWS_PB; ++k) {
            const int r = r0 + k;
            if (r < own_hi) acc += fabs(analytic_at(r, c, N, h, min_x, min_y) - U[(size_t)(r - base) * N + c]);
        }
    }
    const double s = block_sum(acc);
    if (threadIdx.x == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = s;
}

// ---------------------------------------------------------------- synthetic data
__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z *= 0x9E3779B97F4A7C15ull;
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}
__global__ __launch_bounds__(TB) void k_fill_uniform(double *__restrict__ dst, size_t n, uint64_t seed)
{
    for (size_t i = (size_t)blockIdx.x * TB + threadIdx.x; i < n; i += (size_t)gridDim.x * TB)
        dst[i] = (double)(mix64((uint64_t)i + seed) >> 11) * (1.0 / 9007199254740992.0);
}
__global__ __launch_bounds__(TB) void k_checksum(const double *__restrict__ src, size_t n,
                                                 unsigned long long *__restrict__ out)
{
    unsigned long long s0 = 0, s1 = 0;
    for (size_t i = (size_t)blockIdx.x * TB + threadIdx.x; i < n; i += (size_t)gridDim.x * TB) {
        const double v = src[i] + 0.0;  // -0.0 -> +0.0
        const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
        s0 += bits;
        s1 += bits * (2ull * (unsigned long long)i + 1ull);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s0 += __shfl_down(s0, o, 64);
        s1 += __shfl_down(s1, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {  // integer atomics: order-independent, deterministic
        atomicAdd(&out[0], s0);
        atomicAdd(&out[1], s1);
    }
}

// ---------------------------------------------------------------- Gauss-Seidel
// src/MG_solver_CPU.cpp:952-1066.  Colour 0 = (row+col) even (the ieven table
// :973-980), colour 1 = odd (:983-990); update :1020/:1043; err :1051-1059.
__device__ __forceinline__ double gs_update(const double *U, const double *F, int p, int N, double h2)
{
    return 0.25 * (U[p - 1] + U[p + 1] + U[p + N] + U[p - N] - h2 * F[p]);
}

constexpr int GS_MAX_ITER = 50000000;

// whole solve in ONE workgroup: U (and F when it fits) live in LDS, convergence is
// tested on the device every iteration exactly like the reference's while loop.
template <bool F_IN_LDS>
__global__ __launch_bounds__(1024) void k_gs_workgroup(int N, double h2, double inv, double *__restrict__ Ug,
                                                       const double *__restrict__ Fg, double tol,
                                                       int *__restrict__ state)
{
    extern __shared__ __align__(16) double lds[];
    __shared__ double s_err;
    double *U = lds;
    const int n = N * N;
    const double *F = Fg;
    if (F_IN_LDS) {
        double *Fl = lds + n;
        for (int p = threadIdx.x; p < n; p += blockDim.x) Fl[p] = Fg[p];
        F = Fl;
    }
    for (int p = threadIdx.x; p < n; p += blockDim.x) U[p] = 0.0;  // memset(U, 0)  :993
    __syncthreads();

    const double denom = (double)((N - 2) * (N - 2));
    int iterations = 0;
    for (;;) {
        for (int colour = 0; colour < 2; ++colour) {
            for (int p = threadIdx.x; p < n; p += blockDim.x) {
                const int r = p / N, c = p - r * N;
                if (!rim(r, c, N) && ((r + c) & 1) == colour) U[p] = gs_update(U, F, p, N, h2);
            }
            __syncthreads();
        }
        ++iterations;
        double acc = 0.0;
        for (int p = threadIdx.x; p < n; p += blockDim.x) {
            const int r = p / N, c = p - r * N;
            if (!rim(r, c, N))
                acc = acc + fabs(inv * (U[p + N] + U[p - N] + U[p + 1] + U[p - 1] - 4 * U[p]) - F[p]);
        }
        const double s = block_sum(acc);
        if (threadIdx.x == 0) s_err = s / denom;
        __syncthreads();
        const double err = s_err;
        if (!(err > tol) || iterations >= GS_MAX_ITER) break;  // while (err > target_error)
    }
    for (int p = threadIdx.x; p < n; p += blockDim.x) Ug[p] = U[p];
    if (threadIdx.x == 0) {
        state[0] = 1;
        state[1] = iterations;
    }
}

// grids of at most 64 points (N <= 8, the coarsest level of every shipped cycle file): ONE
// wave, one point per lane, U in a register (mg_gs_wave.h) -- no LDS traffic, no barriers.
__global__ __launch_bounds__(64) void k_gs_wave(int N, double h2, double inv, double *__restrict__ Ug,
                                                const double *__restrict__ Fg, double tol, int *__restrict__ state)
{
    const int lane = threadIdx.x;
    const int n = N * N;
    const double f = lane < n ? Fg[lane] : 0.0;
    int iterations = 0;
    const double u = gsw::solve(N, h2, inv, f, tol, GS_MAX_ITER, &iterations);
    if (lane < n) Ug[lane] = u;
    if (lane == 0) {
        state[0] = 1;
        state[1] = iterations;
    }
}

// multi-workgroup form for grids that do not fit one CU's LDS: one launch per colour
// plus a norm; every kernel is a no-op once state[0] (done) is set, so the host may
// enqueue iterations in batches without changing the result.
__global__ __launch_bounds__(TB) void k_gs_colour(int N, double h2, double *__restrict__ U,
                                                  const double *__restrict__ F, int colour,
                                                  const int *__restrict__ state)
{
    if (state[0]) return;
    const int c = blockIdx.x * TB + threadIdx.x, r = blockIdx.y;
    if (c >= N || rim(r, c, N) || ((r + c) & 1) != colour) return;
    const size_t p = (size_t)r * N + c;
    U[p] = 0.25 * (U[p - 1] + U[p + 1] + U[p + N] + U[p - N] - h2 * F[p]);
}
__global__ __launch_bounds__(TB) void k_gs_norm(int N, double inv, const double *__restrict__ U,
                                                const double *__restrict__ F, double *__restrict__ part,
                                                const int *__restrict__ state)
{
    if (state[0]) return;
    const int c = blockIdx.x * TB + threadIdx.x, r = blockIdx.y;
    double acc = 0.0;
    if (c < N && !rim(r, c, N)) {
        const size_t p = (size_t)r * N + c;
        acc = fabs(inv * star_minus4(U, p, N) - F[p]);
    }
    const double s = block_sum(acc);
    if (threadIdx.x == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = s;
}
__global__ __launch_bounds__(1024) void k_gs_check(const double *__restrict__ part, size_t n, int N, double tol,
                                                   int *__restrict__ state)
{
    if (state[0]) return;
    double acc = 0.0;
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) acc += part[i];
    const double s = block_sum(acc);
    if (threadIdx.x == 0) {
        const double err = s / (double)((N - 2) * (N - 2));
        state[1] += 1;
        if (!(err > tol) || state[1] >= GS_MAX_ITER) state[0] = 1;
    }
}

inline dim3 grid_rows(int N, int rows_per_block) { return dim3((N + TB - 1) / TB, (N + rows_per_block - 1) / rows_per_block); }
// operator-by-operator kernels: from this grid size on (arrays far larger than the caches) the 16-byte non-temporal forms
inline int big_grid_min()
{
    static const int v = [] { const char *e = getenv("MG_BIG_GRID_MIN_N"); return e ? atoi(e) : 4096; }();
    return v;
}
inline int grid_flat(size_t n)
{
    size_t b = (n + TB - 1) / TB;
    const size_t cap = 256 * 16;
    return (int)(b < cap ? (b ? b : 1) : cap);
}

void finish(hipStream_t s, const double *part, size_t n, int mode, int N, double *out)
{
    hipLaunchKernelGGL(k_finish, dim3(1), dim3(1024), 0, s, part, n, mode, N, out);
}

}  // namespace

// ------------------------------------------------------------------ launchers
void finish_smoothing_error(hipStream_t s, const double *part, size_t n, int N, double *out)
{
    finish(s, part, n, N > 0 ? FIN_SMOOTH_ERR : FIN_RAW, N, out);
}

void analytic_error_rows(hipStream_t s, int N, double L, const double *U, const RowWindow &w, double min_x,
                         double min_y, double *out_raw)
{
    const int own = w.own_hi - w.own_lo;
    if (own <= 0) {
        (void)hipMemsetAsync(out_raw, 0, sizeof(double), s);
        return;
    }
    const dim3 g((N + TB - 1) / TB, (own + ROWS_PB - 1) / ROWS_PB);
    const size_t np = (size_t)g.x * g.y;
    double *part = partials(np);
    hipLaunchKernelGGL(k_analytic_error_rows, g, dim3(TB), 0, s, N, L / (double)(N - 1), U, w.base, w.own_lo, min_x, min_y,
                       w.own_hi, part);
    finish(s, part, np, FIN_RAW, N, out_raw);
}

void finish_smoothing_errors(hipStream_t s, const NormBatch &b, int count)
{
    if (count > 0) hipLaunchKernelGGL(k_finish_batch, dim3(count), dim3(1024), 0, s, b);
}

void jacobi_simple(hipStream_t s, int N, double dx2, const double *in, const double *F, double *out)
{
    const dim3 g = grid_rows(N, ROWS_PB);
    if (in && N % 2 == 0 && N >= 512) {
        // (measured, scripts/perf_pair.py: N = 8192 321 -> 301 us = 0.63 -> 0.67 of the HBM peak, 16384 1215 -> 1130 us = 0.71; at 4096,
        // where the arrays sit in the caches, the one-row form is faster: 62 against 73 us; 8 rows per thread lose at 8192: 358 us)
        static const int pair_rows_min = [] { const char *e = getenv("MG_PAIR_ROWS_MIN_N"); return e ? atoi(e) : 8192; }();
        if (N >= pair_rows_min) {
            hipLaunchKernelGGL(k_jacobi_pair_rows, dim3((N / 2 + TB - 1) / TB, (N + PR - 1) / PR), dim3(TB), 0, s, N, dx2, in, F, out);
            return;
        }
        hipLaunchKernelGGL(k_jacobi_pair, dim3((N / 2 + TB - 1) / TB, N), dim3(TB), 0, s, N, dx2, in, F, out);
        return;
    }
    if (in) hipLaunchKernelGGL(k_jacobi_simple<false>, g, dim3(TB), 0, s, N, dx2, in, F, out);
    else hipLaunchKernelGGL(k_jacobi_simple<true>, g, dim3(TB), 0, s, N, dx2, in, F, out);
}

void residual(hipStream_t s, int N, double inv, const double *U, const double *F, double *D, int sign)
{
    if (N % 2 == 0 && N >= big_grid_min()) {
        hipLaunchKernelGGL(k_residual_pairs, dim3((N / 2 + TB - 1) / TB, (N + PR - 1) / PR), dim3(TB), 0, s, N, inv, U, F, D, sign);
        return;
    }
    hipLaunchKernelGGL(k_residual, grid_rows(N, ROWS_PB), dim3(TB), 0, s, N, inv, U, F, D, sign);
}

void smoothing_error(hipStream_t s, int N, double inv, const double *U, const double *F, double *out)
{
    const dim3 g = grid_rows(N, ROWS_PB);
    const size_t np = (size_t)g.x * g.y;
    double *part = partials(np);
    hipLaunchKernelGGL(k_smoothing_error, g, dim3(TB), 0, s, N, inv, U, F, part);
    finish(s, part, np, FIN_SMOOTH_ERR, N, out);
}

void restrict_gather(hipStream_t s, int N, const double *Uf, int M, double *Uc, const RestrictTable &t, int sign)
{
    hipLaunchKernelGGL(k_restrict<double>, dim3((M + TB - 1) / TB, M), dim3(TB), 0, s, N, Uf, M, Uc, t.lo, t.w, sign);
}
void restrict_gather_f32(hipStream_t s, int N, const float *Uf, int M, float *Uc, const RestrictTable &t, int sign)
{
    hipLaunchKernelGGL(k_restrict<float>, dim3((M + TB - 1) / TB, M), dim3(TB), 0, s, N, Uf, M, Uc, t.lo, t.w_f, sign);
}

void prolong(hipStream_t s, int N, const double *Uc, int M, const double *Uf_in, double *Uf_out, const ProlongTable &t)
{
    if (M % 2 == 0 && M >= big_grid_min() && t.fusable) {
        const dim3 g2((M / 2 + TB - 1) / TB, M);
        const double rc = 1.0 / t.c_dx;   // IEEE division on the host: correctly rounded
        if (Uf_in)
            hipLaunchKernelGGL((k_prolong_pairs<true>), g2, dim3(TB), 0, s, N, Uc, M, Uf_in, Uf_out, t.owner_row, t.owner_col, t.row_hi, t.row_lo,
                               t.col_hi, t.col_lo, t.c_dx, rc);
        else
            hipLaunchKernelGGL((k_prolong_pairs<false>), g2, dim3(TB), 0, s, N, Uc, M, Uf_in, Uf_out, t.owner_row, t.owner_col, t.row_hi, t.row_lo,
                               t.col_hi, t.col_lo, t.c_dx, rc);
        return;
    }
    const dim3 g((M + TB - 1) / TB, M);
    if (Uf_in)
        hipLaunchKernelGGL((k_prolong<true, double>), g, dim3(TB), 0, s, N, Uc, M, Uf_in, Uf_out, t.owner_row, t.owner_col,
                           t.row_hi, t.row_lo, t.col_hi, t.col_lo, t.c_dx);
    else
        hipLaunchKernelGGL((k_prolong<false, double>), g, dim3(TB), 0, s, N, Uc, M, Uf_in, Uf_out, t.owner_row, t.owner_col,
                           t.row_hi, t.row_lo, t.col_hi, t.col_lo, t.c_dx);
}
// U_out = U_in + P(U_c) on fp32 fields (weights rounded from the fp64 tables, fp32 divisions)
void prolong_add_f32(hipStream_t s, int N, const float *Uc, int M, const float *Uf_in, float *Uf_out, const ProlongTable &t)
{
    const dim3 g((M + TB - 1) / TB, M);
    hipLaunchKernelGGL((k_prolong<true, float>), g, dim3(TB), 0, s, N, Uc, M, Uf_in, Uf_out, t.owner_row, t.owner_col,
                       t.row_hi_f, t.row_lo_f, t.col_hi_f, t.col_lo_f, (float)t.c_dx);
}

void convert_to_f32(hipStream_t s, float *dst, const double *src, size_t n)
{
    hipLaunchKernelGGL(k_to_f32, dim3(grid_flat(n)), dim3(TB), 0, s, dst, src, n);
}
void convert_to_f64(hipStream_t s, double *dst, const float *src, size_t n)
{
    hipLaunchKernelGGL(k_to_f64, dim3(grid_flat(n)), dim3(TB), 0, s, dst, src, n);
}

void refine_residual(hipStream_t s, int N, double inv, const double *U, const double *F, float *src, double *err_out)
{
    if (N % 2 == 0 && N >= 1024) {   // 16 B per lane, rolling row window (same expressions, same bits)
        const dim3 g((N / 2 + TB - 1) / TB, (N + RR - 1) / RR);
        const size_t np = (size_t)g.x * g.y;
        double *part = partials(np);
        hipLaunchKernelGGL(k_refine_residual_pairs, g, dim3(TB), 0, s, N, inv, U, F, src, part, 0, 0, N);
        finish(s, part, np, FIN_SMOOTH_ERR, N, err_out);
        return;
    }
    const dim3 g = grid_rows(N, ROWS_PB);
    const size_t np = (size_t)g.x * g.y;
    double *part = partials(np);
    hipLaunchKernelGGL(k_refine_residual, g, dim3(TB), 0, s, N, inv, U, F, src, part);
    finish(s, part, np, FIN_SMOOTH_ERR, N, err_out);
}
void refine_residual_rows(hipStream_t s, int N, double inv, const double *U, const double *F, float *src, const RowWindow &w,
                          double *out_raw)
{
    const int own = w.own_hi - w.own_lo;
    if (own <= 0) {
        (void)hipMemsetAsync(out_raw, 0, sizeof(double), s);
        return;
    }
    if (N % 2 == 0 && N >= 1024) {   // 16 B per lane, rolling row window (same expressions, same bits)
        const dim3 g((N / 2 + TB - 1) / TB, (own + RR - 1) / RR);
        const size_t np = (size_t)g.x * g.y;
        double *part = partials(np);
        hipLaunchKernelGGL(k_refine_residual_pairs, g, dim3(TB), 0, s, N, inv, U, F, src, part, w.base, w.own_lo, w.own_hi);
        finish(s, part, np, FIN_RAW, N, out_raw);
        return;
    }
    const dim3 g((N + TB - 1) / TB, (own + ROWS_PB - 1) / ROWS_PB);
    const size_t np = (size_t)g.x * g.y;
    double *part = partials(np);
    hipLaunchKernelGGL(k_refine_residual_rows, g, dim3(TB), 0, s, N, inv, U, F, src, w.base, w.own_lo, w.own_hi, part);
    finish(s, part, np, FIN_RAW, N, out_raw);
}
void add_widened(hipStream_t s, double *U, const float *e, size_t n)
{
    if (n % 2 == 0 && n >= ((size_t)1 << 20) && ((uintptr_t)U % 16) == 0 && ((uintptr_t)e % 8) == 0) {
        const size_t n2 = n / 2;
        hipLaunchKernelGGL(k_add_widened_pairs, dim3((unsigned)((n2 + TB - 1) / TB)), dim3(TB), 0, s, U, e, n2);
        return;
    }
    hipLaunchKernelGGL(k_add_widened, dim3(grid_flat(n)), dim3(TB), 0, s, U, e, n);
}

void add(hipStream_t s, size_t n, double *a, const double *b)
{
    if (n % 2 == 0 && n >= (size_t)big_grid_min() * big_grid_min() && ((uintptr_t)a % 16) == 0 && ((uintptr_t)b % 16) == 0) {
        hipLaunchKernelGGL(k_add_pairs, dim3((unsigned)((n / 2 + TB - 1) / TB)), dim3(TB), 0, s, n / 2, a, b);
        return;
    }
    hipLaunchKernelGGL(k_add, dim3(grid_flat(n)), dim3(TB), 0, s, n, a, b);
}
void negate(hipStream_t s, size_t n, double *a)
{
    if (n % 2 == 0 && n >= (size_t)big_grid_min() * big_grid_min() && ((uintptr_t)a % 16) == 0) {
        hipLaunchKernelGGL(k_negate_pairs, dim3((unsigned)((n / 2 + TB - 1) / TB)), dim3(TB), 0, s, n / 2, a);
        return;
    }
    hipLaunchKernelGGL(k_negate, dim3(grid_flat(n)), dim3(TB), 0, s, n, a);
}

void source_device(hipStream_t s, int N, double L, double *F, double min_x, double min_y, int row_lo, int row_hi)
{
    if (row_hi <= row_lo) return;
    hipLaunchKernelGGL(k_source, dim3((N + TB - 1) / TB, row_hi - row_lo), dim3(TB), 0, s, N, L / (double)(N - 1), F, min_x, min_y,
                       row_lo);
}
void analytic(hipStream_t s, int N, double L, double *U, double min_x, double min_y)
{
    hipLaunchKernelGGL(k_analytic, dim3((N + TB - 1) / TB, N), dim3(TB), 0, s, N, L / (double)(N - 1), U, min_x, min_y);
}
void analytic_error(hipStream_t s, int N, double L, const double *U, double min_x, double min_y, double *out)
{
    const dim3 g = grid_rows(N, ROWS_PB);
    const size_t np = (size_t)g.x * g.y;
    double *part = partials(np);
    hipLaunchKernelGGL(k_analytic_error, g, dim3(TB), 0, s, N, L / (double)(N - 1), U, min_x, min_y, part);
    finish(s, part, np, FIN_MEAN_NN, N, out);
}

void fill_uniform(hipStream_t s, double *dst, size_t n, uint64_t seed)
{
    hipLaunchKernelGGL(k_fill_uniform, dim3(grid_flat(n)), dim3(TB), 0, s, dst, n, seed);
}
void checksum(hipStream_t s, const double *src, size_t n, uint64_t *out_dev)
{
    (void)hipMemsetAsync(out_dev, 0, 2 * sizeof(uint64_t), s);
    hipLaunchKernelGGL(k_checksum, dim3(grid_flat(n)), dim3(TB), 0, s, src, n, (unsigned long long *)out_dev);
}

int gs_single_workgroup_max_n() { return 128; }

void gauss_seidel(hipStream_t s, int N, double h2, double inv, double *U, const double *F, double tol, int *state)
{
    (void)hipMemsetAsync(state, 0, 4 * sizeof(int), s);
    const size_t n = (size_t)N * N;
    if (n <= 64 && N >= 4 && (N & 1) == 0) {  // 4 x 4, 6 x 6, 8 x 8: the block solver of the coarse tail
        gauss_seidel_blocks_launch(s, N, h2, inv, U, F, tol, state);
        return;
    }
    if (n <= 64) {
        hipLaunchKernelGGL(k_gs_wave, dim3(1), dim3(64), 0, s, N, h2, inv, U, F, tol, state);
        return;
    }
    if (N <= gs_single_workgroup_max_n()) {
        const bool f_in_lds = N <= 96;
        const size_t lds = n * sizeof(double) * (f_in_lds ? 2 : 1);
        int threads = (int)((n + 63) / 64 * 64);
        if (threads > 1024) threads = 1024;
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void *)k_gs_workgroup<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
            (void)hipFuncSetAttribute((const void *)k_gs_workgroup<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
            attr_set = true;
        }
        if (f_in_lds) hipLaunchKernelGGL(k_gs_workgroup<true>, dim3(1), dim3(threads), lds, s, N, h2, inv, U, F, tol, state);
        else hipLaunchKernelGGL(k_gs_workgroup<false>, dim3(1), dim3(threads), lds, s, N, h2, inv, U, F, tol, state);
        return;
    }
    // large coarsest grids: batches of iterations, host polls the done flag
    (void)hipMemsetAsync(U, 0, n * sizeof(double), s);
    const dim3 g((N + TB - 1) / TB, N);
    const size_t np = (size_t)g.x * g.y;
    double *part = partials(np);
    Context &c = ctx();
    for (;;) {
        for (int it = 0; it < 32; ++it) {
            hipLaunchKernelGGL(k_gs_colour, g, dim3(TB), 0, s, N, h2, U, F, 0, state);
            hipLaunchKernelGGL(k_gs_colour, g, dim3(TB), 0, s, N, h2, U, F, 1, state);
            hipLaunchKernelGGL(k_gs_norm, g, dim3(TB), 0, s, N, inv, U, F, part, state);
            hipLaunchKernelGGL(k_gs_check, dim3(1), dim3(1024), 0, s, part, np, N, tol, state);
        }
        if (!MG_HIP(hipMemcpyAsync(c.host_ints, state, 2 * sizeof(int), hipMemcpyDeviceToHost, s))) return;
        if (!MG_HIP(hipStreamSynchronize(s))) return;
        if (c.host_ints[0]) break;
    }
}

}  // namespace k
}  // namespace mg
