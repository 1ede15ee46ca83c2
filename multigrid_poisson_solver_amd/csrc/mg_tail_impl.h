// mg_tail_impl.h -- the coarse tail of a cycle in ONE launch.
//
// Below N = 64 a level is a few KiB: every operator launch is pure latency (one wave per
// SIMD, ~12 us per fused node, SURVEY.md section 7 hard part 6) and a W-cycle visits these
// levels hundreds of times.  This kernel keeps U, its ping-pong partner and F of ALL levels
// N <= 64 in the LDS of one workgroup (3*8*(64^2+32^2+16^2+8^2) = 127.5 KiB of the 160 KiB)
// and interprets the slice of the cycle-structure node stream that stays on those levels
// (-1: smooth+restrict, 0: red-black Gauss-Seidel, 1: prolong+add+smooth) without leaving
// the CU.  Every expression is the one the per-operator kernels use (reference order,
// -ffp-contract=off), so the result is bit-identical to running the nodes one by one.
//
// Kernel source for both field types (MG_REAL = double: mg_tail.hip; float: mg_tail_f32.hip, the
// mixed-precision mode); with MG_REAL = double every expression is what it was before the split.
#include <hip/hip_runtime.h>

#include "mg_gs_wave.h"
#include "mg_internal.h"

#if !defined(MG_REAL) || !defined(MG_REAL_NS)
#error "define MG_REAL (double|float) and MG_REAL_NS (f64|f32) before including mg_tail_impl.h"
#endif

namespace mg {
namespace k {
namespace MG_REAL_NS {

typedef MG_REAL real_t;


constexpr int TAIL_THREADS = 1024;
constexpr int TAIL_WAVES = TAIL_THREADS / 64;

// every point of an N x N LDS grid, rows over waves, columns over lanes: no integer division
#define FOR_POINTS(N, r, c, p)                                             \
    for (int r = (int)(threadIdx.x >> 6); r < (N); r += TAIL_WAVES)        \
        for (int c = (int)(threadIdx.x & 63), p = r * (N) + c; c < (N); c += 64, p += 64)

__device__ __forceinline__ bool rim(int r, int c, int N) { return r == 0 || c == 0 || r == N - 1 || c == N - 1; }

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
// fixed-order block sum, result broadcast to every thread
__device__ __forceinline__ double block_total(double v, double *sm /* [17] */)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) sm[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0.0;
        for (int i = 0; i < TAIL_THREADS / 64; ++i) r += sm[i];
        sm[16] = r;
    }
    __syncthreads();
    return sm[16];
}

// All level arrays live in ONE dynamic LDS array and are addressed by integer offsets, so every
// access is a ds_read/ds_write (pointers into LDS stored in variables decay to flat accesses).
extern __shared__ __align__(16) real_t lds[];
#define SRC(i) lds[src + (i)]
#define FF(i) lds[F + (i)]

// one Jacobi sweep src -> dst (src/MG_solver_CPU.cpp:587-599), rim keeps its value
__device__ void sweep(int N, real_t dx2, int src, int F, int dst)
{
    FOR_POINTS(N, r, c, p)
    {
        real_t v = SRC(p);
        if (!rim(r, c, N)) v = v + real_t(0.25) * (SRC(p + N) + SRC(p - N) + SRC(p + 1) + SRC(p - 1) - 4 * SRC(p) - dx2 * FF(p));
        lds[dst + p] = v;
    }
    __syncthreads();
}

// doSmoothing's error (:607-622)
__device__ double smoothing_error(int N, real_t inv, int src, int F, double *sm)
{
    double acc = 0.0;
    FOR_POINTS(N, r, c, p)
    {
        if (!rim(r, c, N) && ((r + c) & 1) == 0)
            acc += fabs((double)(inv * (SRC(p + N) + SRC(p - N) + SRC(p + 1) + SRC(p - 1) - 4 * SRC(p)) - FF(p)));
    }
    const double s = block_total(acc, sm);
    double e = s + s;
    e = e / N / N;
    return e;
}

// signed residual at one fine point: -(getResidual) as the driver forms it (:268, :277-280)
__device__ __forceinline__ real_t neg_residual(int N, real_t inv, int src, int F, int r, int c)
{
    const int q = r * N + c;
    real_t v = 0.0;
    if (!rim(r, c, N)) v = inv * (SRC(q + N) + SRC(q - N) + SRC(q + 1) + SRC(q - 1) - 4 * SRC(q)) - FF(q);
    return -v;
}

// The exact solver always computes in fp64, also under fp32 fields (mixed-precision mode): its
// tolerance is the error floor of the whole cycle and fp32 round-off cannot reach the reference's
// 1e-7.  F is widened (exactly), the reference's fp64 iteration runs unchanged, U is rounded once.
// GSD(o, i): element i of a double array that starts `o` doubles into the LDS array.
#define GSD(o, i) (reinterpret_cast<double *>(lds)[(o) + (i)])

// red-black Gauss-Seidel (:952-1066) on fp64 LDS arrays u / f, all threads
__device__ void gauss_seidel_block(int N, double h2, double inv, int u, int f, double tol, double *sm, int *state)
{
    FOR_POINTS(N, r, c, p) GSD(u, p) = 0.0;  // :993
    __syncthreads();
    const double denom = (double)((N - 2) * (N - 2));
    int iterations = 0;
    for (;;) {
        for (int colour = 0; colour < 2; ++colour) {
            FOR_POINTS(N, r, c, p)
            {
                if (!rim(r, c, N) && ((r + c) & 1) == colour)
                    GSD(u, p) = 0.25 * (GSD(u, p - 1) + GSD(u, p + 1) + GSD(u, p + N) + GSD(u, p - N) - h2 * GSD(f, p));  // :1020
            }
            __syncthreads();
        }
        ++iterations;
        double acc = 0.0;
        FOR_POINTS(N, r, c, p)
        {
            if (!rim(r, c, N))
                acc = acc + fabs(inv * (GSD(u, p + N) + GSD(u, p - N) + GSD(u, p + 1) + GSD(u, p - 1) - 4 * GSD(u, p)) - GSD(f, p));
        }
        const double err = block_total(acc, sm) / denom;  // :1059
        if (!(err > tol) || iterations >= 50000000) break;
    }
    if (threadIdx.x == 0) {
        state[0] = 1;
        state[1] = iterations;
    }
}

// level arrays src / F are real_t; `scratch` = offset (in doubles) of 2*N*N spare doubles, used
// only when real_t is not double
__device__ void gauss_seidel_level(int N, double h2, double inv, int src, int F, int scratch, double tol, double *sm, int *state)
{
    if (sizeof(real_t) == sizeof(double)) {
        gauss_seidel_block(N, h2, inv, src, F, tol, sm, state);
    } else {
        const int u = scratch, f = scratch + N * N;
        FOR_POINTS(N, r, c, p) GSD(f, p) = (double)FF(p);
        __syncthreads();
        gauss_seidel_block(N, h2, inv, u, f, tol, sm, state);
        __syncthreads();
        FOR_POINTS(N, r, c, p) SRC(p) = (real_t)GSD(u, p);
    }
}

// the same solve for a grid of at most 64 points, by wave 0 alone with U in registers
// (mg_gs_wave.h); the other waves wait at the barrier
__device__ void gauss_seidel_wave(int N, double h2, double inv, int src, int F, double tol, int *state)
{
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const int n = N * N;
        const double f = lane < n ? (double)FF(lane) : 0.0;
        int iterations = 0;
        const double u = gsw::solve<double>(N, h2, inv, f, tol, 50000000, &iterations);
        if (lane < n) SRC(lane) = (real_t)u;
        if (lane == 0) {
            state[0] = 1;
            state[1] = iterations;
        }
    }
    __syncthreads();
}

// offset of level l's three arrays (U, partner, F) in the LDS array
__device__ __forceinline__ int level_base(const TailArgsT<real_t> &a, int l)
{
    int o = 0;
#pragma unroll
    for (int k = 0; k < TAIL_MAX_LEVELS; ++k)
        if (k < l) o += 3 * a.N[k] * a.N[k];
    return o;
}

// first double past the level arrays (fp32 fields only: fp64 scratch of the exact solver)
__host__ __device__ __forceinline__ int gs_scratch(const TailArgsT<real_t> &a)
{
    size_t elems = 0;
    for (int l = 0; l < a.n_levels; ++l) elems += (size_t)3 * a.N[l] * a.N[l];
    return (int)((elems * sizeof(real_t) + 7) / 8);
}

__global__ __launch_bounds__(TAIL_THREADS) void k_tail(const TailArgsT<real_t> a)
{
    __shared__ double sm[17];
    unsigned swapped = 0;  // bit l: level l's U currently lives in its second buffer (same in every thread)
    auto U_of = [&](int l) { return level_base(a, l) + (((swapped >> l) & 1u) ? a.N[l] * a.N[l] : 0); };
    auto T_of = [&](int l) { return level_base(a, l) + (((swapped >> l) & 1u) ? 0 : a.N[l] * a.N[l]); };
    auto F_of = [&](int l) { return level_base(a, l) + 2 * a.N[l] * a.N[l]; };
    {
        const int N0 = a.N[0], f0 = F_of(0);
        FOR_POINTS(N0, r, c, p) lds[f0 + p] = a.F_top[p];
    }
    __syncthreads();

    int cur = 0;
    for (int i = 0; i < a.n_nodes; ++i) {
        const TailNode nd = a.nodes[i];
        if (nd.type == -1) {
            // memset(U,0) :256, doSmoothing :259, getResidual :268, sign flip :277-280, doRestriction :287
            const int N = a.N[cur], F = F_of(cur);
            {
                const int u = U_of(cur);
                FOR_POINTS(N, r, c, p) lds[u + p] = 0.0;
            }
            __syncthreads();
            for (int s = 0; s < nd.steps; ++s) {
                sweep(N, a.dx2[cur], U_of(cur), F, T_of(cur));
                swapped ^= 1u << cur;
            }
            const int src = U_of(cur);
            const double e = smoothing_error(N, a.inv[cur], src, F, sm);
            if (threadIdx.x == 0 && nd.err_slot >= 0) a.err_dev[nd.err_slot] = e;
            const int M = a.N[cur + 1], Fc = F_of(cur + 1);
            const int *lo = a.r_lo[cur];
            const real_t *w = a.r_w[cur];
            const real_t inv = a.inv[cur];
            FOR_POINTS(M, rc, cc, q)
            {
                real_t v = 0.0;
                if (!rim(rc, cc, M)) {
                    const real_t wa = w[cc], wb = real_t(1.0) - wa, wc = w[rc], wd = real_t(1.0) - wc;
                    const int fr = lo[rc], fc = lo[cc];
                    const real_t u0 = neg_residual(N, inv, src, F, fr, fc);
                    const real_t u1 = neg_residual(N, inv, src, F, fr, fc + 1);
                    const real_t u2 = neg_residual(N, inv, src, F, fr + 1, fc);
                    const real_t u3 = neg_residual(N, inv, src, F, fr + 1, fc + 1);
                    v = wb * wd * u0 + wa * wd * u1 + wc * wb * u2 + wa * wc * u3;  // :676
                }
                lds[Fc + q] = v;
            }
            __syncthreads();
            ++cur;
        } else if (nd.type == 0) {
            const int N = a.N[cur];
            if (N * N <= 64) gauss_seidel_wave(N, a.gs_h2[cur], a.gs_inv[cur], U_of(cur), F_of(cur), nd.tol, a.gs_state);
            else gauss_seidel_level(N, a.gs_h2[cur], a.gs_inv[cur], U_of(cur), F_of(cur), gs_scratch(a), nd.tol, sm, a.gs_state);
            __syncthreads();
        } else {  // 1: doProlongation :354, doGridAddition :368, doSmoothing :416
            const int Nc = a.N[cur], fine = cur - 1, N = a.N[fine];
            const int uc = U_of(cur), uf = U_of(fine), F = F_of(fine);
            const int *orow = a.p_orow[fine], *ocol = a.p_ocol[fine];
            const real_t *rhi = a.p_rhi[fine], *rlo = a.p_rlo[fine], *chi = a.p_chi[fine], *clo = a.p_clo[fine];
            const real_t c_dx = a.c_dx[fine];
            FOR_POINTS(N, kf, l, q)
            {
                const int ci = orow[kf], cj = ocol[l];
                if (ci < 0 || cj < 0) continue;
                const int p = uc + ci * Nc + cj;
                const real_t c1 = lds[p], c2 = lds[p + 1], c3 = lds[p + Nc], c4 = lds[p + Nc + 1];
                const real_t v = ((c1 * chi[l] + c2 * clo[l]) * rhi[kf] + (c3 * chi[l] + c4 * clo[l]) * rlo[kf]) / c_dx / c_dx;
                lds[uf + q] = lds[uf + q] + v;
            }
            __syncthreads();
            for (int s = 0; s < nd.steps; ++s) {
                sweep(N, a.dx2[fine], U_of(fine), F, T_of(fine));
                swapped ^= 1u << fine;
            }
            const double e = smoothing_error(N, a.inv[fine], U_of(fine), F, sm);
            if (threadIdx.x == 0 && nd.err_slot >= 0) a.err_dev[nd.err_slot] = e;
            --cur;
        }
    }
    {
        const int N0 = a.N[0], u0 = U_of(0);
        FOR_POINTS(N0, r, c, p) a.U_top[p] = lds[u0 + p];
    }
}


inline size_t tail_lds_bytes(const TailArgsT<real_t> &a)
{
    size_t bytes = 0;
    for (int l = 0; l < a.n_levels; ++l) bytes += (size_t)3 * a.N[l] * a.N[l] * sizeof(real_t);
    if (sizeof(real_t) != sizeof(double)) {
        // fp64 scratch (U, F) for exact-solver nodes on levels too big for the one-wave solver
        int depth = 0, worst = 0;
        for (int i = 0; i < a.n_nodes; ++i) {
            if (a.nodes[i].type == -1) ++depth;
            else if (a.nodes[i].type == 1) --depth;
            else if (a.N[depth] * a.N[depth] > 64 && a.N[depth] > worst) worst = a.N[depth];
        }
        if (worst) bytes = (size_t)gs_scratch(a) * 8 + (size_t)2 * worst * worst * sizeof(double);
    }
    return bytes;
}

inline void tail_launch(hipStream_t s, const TailArgsT<real_t> &a)
{
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)k_tail, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
        attr_set = true;
    }
    hipLaunchKernelGGL(k_tail, dim3(1), dim3(TAIL_THREADS), tail_lds_bytes(a), s, a);
}

}  // namespace MG_REAL_NS
}  // namespace k
}  // namespace mg
