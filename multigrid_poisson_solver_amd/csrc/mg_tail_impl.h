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
// One CU issues 64 lane-instructions per clock, so the kernel is bound by its instruction count,
// not by LDS or HBM: a thread therefore owns FIXED points of a level (flat index, up to 4 at
// N = 64) and keeps their U, F and h^2*F in registers across the sweeps of a node (4 LDS reads and
// 1 write per point and sweep), the first sweep of a `-1` node starts from the zero field without
// touching LDS, the residual is formed once per point (error norm and restriction share it), the
// transfer tables are staged in LDS once per launch, `/c_dx/c_dx` is the correctly rounded
// division by a constant of the streaming kernel, and the norm's last stage is left to thread 0
// while the other waves go on.
//
// Kernel source for both field types (MG_REAL = double: mg_tail.hip; float: mg_tail_f32.hip, the
// mixed-precision mode); with MG_REAL = double every expression is what it was before the split.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "mg_divconst.h"
#include "mg_gs_wave.h"
#include "mg_internal.h"

#if !defined(MG_REAL) || !defined(MG_REAL_NS)
#error "define MG_REAL (double|float) and MG_REAL_NS (f64|f32) before including mg_tail_impl.h"
#endif

namespace mg {
namespace k {
namespace MG_REAL_NS {

typedef MG_REAL real_t;


#ifndef MG_TAIL_THREADS
#define MG_TAIL_THREADS 1024
#endif
constexpr int TAIL_THREADS = MG_TAIL_THREADS;
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

// ---------------------------------------------------------------------------------------------
// flat point ownership: thread t owns points t, t + 1024, ... of the level being worked on
constexpr int PT = (TAIL_MAX_N * TAIL_MAX_N + TAIL_THREADS - 1) / TAIL_THREADS;

// NP = points per thread the code is instantiated for: 1 up to 32 x 32, 2 up to 45 x 45, 4 up to 64 x 64 (a level with
// one point per thread must not pay the predicated second, third and fourth iterations of every loop: the kernel is
// bound by its instruction count)
template <int NP>
struct PointsT {
    int p[NP], r[NP], c[NP];
    bool live[NP], inner[NP], even[NP];
};
template <int NP>
__device__ __forceinline__ PointsT<NP> map_points(int N)
{
    PointsT<NP> P;
    // row = floor(p / N) through one fp32 multiplication: (p + 0.5)/N is at least 0.5/N away from
    // every integer, orders of magnitude more than the rounding error at p < 4096
    const float rn = 1.0f / (float)N;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const int p = (int)threadIdx.x + k * TAIL_THREADS;
        const int r = (int)(((float)p + 0.5f) * rn), c = p - r * N;
        P.p[k] = p;
        P.r[k] = r;
        P.c[k] = c;
        P.live[k] = p < N * N;
        P.inner[k] = P.live[k] && r > 0 && c > 0 && r < N - 1 && c < N - 1;
        P.even[k] = ((r + c) & 1) == 0;
    }
    return P;
}

// one Jacobi sweep src -> dst (src/MG_solver_CPU.cpp:587-599); v = this thread's own points of
// src (kept in registers), rim keeps its value.  ZERO: src is the zero field (:256), nothing is read.
template <bool ZERO, int NP>
__device__ __forceinline__ void sweep(const PointsT<NP> &P, int N, int src, int dst, real_t (&v)[NP], const real_t (&h2f)[NP])
{
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        if (!P.live[k]) continue;
        const int p = P.p[k];
        real_t nv = v[k];
        if (P.inner[k]) {
            if (ZERO) {
                const real_t z = 0.0;
                nv = z + real_t(0.25) * (z + z + z + z - 4 * z - h2f[k]);
            } else {
                nv = v[k] + real_t(0.25) * (SRC(p + N) + SRC(p - N) + SRC(p + 1) + SRC(p - 1) - 4 * v[k] - h2f[k]);
            }
        }
        lds[dst + p] = nv;
        v[k] = nv;
    }
    __syncthreads();
}

// first stage of doSmoothing's error (:607-622): this wave's sum into its slot
__device__ __forceinline__ void post_partial(double acc, double *slots)
{
    const double t = gsw::wave_total(acc);  // all 64 lanes are here
    if ((threadIdx.x & 63) == 0) slots[threadIdx.x >> 6] = t;
}
// last stage, one thread, after a barrier: fixed-order sum, (sum1 + sum2)/N/N  (:621-622); a division by a power of
// two is a change of exponent (v_ldexp_f64; like the division, correctly rounded should the result be subnormal)
__device__ __forceinline__ void finish_error(const double *slots, int N, double *out)
{
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < TAIL_WAVES; ++i) s += slots[i];
    double e = s + s;
    if ((N & (N - 1)) == 0) e = ldexp(e, -2 * (31 - __builtin_clz(N)));
    else e = e / N / N;
    *out = e;
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

// The same solve for a grid of 65..256 points (N = 9..16: the coarsest level of hierarchies whose size is not
// a power of two, e.g. 23168 -> ... -> 11) by wave 0 alone, up to four points per lane, U in an fp64 LDS array.
// One wave needs no barrier: its LDS operations complete in order, and within a colour pass the points that
// are written and the points that are read have different colours.  (The all-thread version above pays five
// 16-wave barriers per iteration: ~1.5 us against ~0.4 us here.)
constexpr int GS_WAVE_PTS = 4;
__device__ void gauss_seidel_wave_lds(int N, double h2, double inv, int u, bool in_place, int src, int F, double tol, int *state)
{
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x, n = N * N;
        const float rn = 1.0f / (float)N;
        int q[GS_WAVE_PTS];
        bool live[GS_WAVE_PTS], inner[GS_WAVE_PTS], red[GS_WAVE_PTS];
        double f[GS_WAVE_PTS], h2f[GS_WAVE_PTS];
#pragma unroll
        for (int k = 0; k < GS_WAVE_PTS; ++k) {
            const int p = lane + 64 * k;
            const int r = (int)(((float)p + 0.5f) * rn), c = p - r * N;  // see map_points
            q[k] = p;
            live[k] = p < n;
            inner[k] = live[k] && r > 0 && c > 0 && r < N - 1 && c < N - 1;
            red[k] = ((r + c) & 1) == 0;
            f[k] = live[k] ? (double)FF(p) : 0.0;
            h2f[k] = h2 * f[k];
            if (live[k]) GSD(u, p) = 0.0;  // memset(U, 0)  :993
        }
        const double denom = (double)((N - 2) * (N - 2));
        // `sum/denom > tol` without the division wherever rounding cannot matter (see gsw::solve)
        const double thr = tol * denom, thr_hi = thr * (1.0 + 0x1p-48), thr_lo = thr * (1.0 - 0x1p-48);
        // own value and the four neighbours of every point stay in registers between the passes: the read
        // that closes an iteration (all neighbours of the final iterate, for the norm) is also the read the
        // next red pass needs, so an iteration is two LDS round trips (black pass, closing read), not three
        double own[GS_WAVE_PTS], nw[GS_WAVE_PTS], ne[GS_WAVE_PTS], nn[GS_WAVE_PTS], ns[GS_WAVE_PTS];
#pragma unroll
        for (int k = 0; k < GS_WAVE_PTS; ++k) own[k] = nw[k] = ne[k] = nn[k] = ns[k] = 0.0;
        int iterations = 0;
        for (;;) {
#pragma unroll
            for (int k = 0; k < GS_WAVE_PTS; ++k) {  // red :1020 -- neighbours from the closing read
                if (inner[k] && red[k]) {
                    own[k] = 0.25 * (nw[k] + ne[k] + nn[k] + ns[k] - h2f[k]);
                    GSD(u, q[k]) = own[k];
                }
            }
#pragma unroll
            for (int k = 0; k < GS_WAVE_PTS; ++k) {  // black :1043
                if (inner[k] && !red[k]) {
                    const int p = q[k];
                    own[k] = 0.25 * (GSD(u, p - 1) + GSD(u, p + 1) + GSD(u, p + N) + GSD(u, p - N) - h2f[k]);
                    GSD(u, p) = own[k];
                }
            }
            ++iterations;
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < GS_WAVE_PTS; ++k) {
                if (inner[k]) {
                    const int p = q[k];
                    nw[k] = GSD(u, p - 1);
                    ne[k] = GSD(u, p + 1);
                    nn[k] = GSD(u, p + N);
                    ns[k] = GSD(u, p - N);
                    const double rs = inv * (nn[k] + ns[k] + ne[k] + nw[k] - 4 * own[k]) - f[k];  // :560
                    acc += fabs(rs);
                }
            }
            const double sum = gsw::wave_total(acc);
            const bool above = sum > thr_hi ? true : (sum < thr_lo ? false : sum / denom > tol);  // :1059, :996
            if (!above || iterations >= 50000000) break;
        }
        if (!in_place) {
#pragma unroll
            for (int k = 0; k < GS_WAVE_PTS; ++k)
                if (live[k]) SRC(q[k]) = (real_t)GSD(u, q[k]);
        }
        if (lane == 0) {
            state[0] = 1;
            state[1] = iterations;
        }
    }
    __syncthreads();
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

// ---------------------------------------------------------------------------------------------
// The same solve for the 8 x 8 coarsest grid of every 2^k hierarchy (and the 4 x 4 / 6 x 6 ones) by wave 0 alone,
// WITHOUT the LDS crossbar and without a full norm in most iterations.  A W-cycle runs this solve hundreds of times
// (512 at N = 8192) at ~76 iterations each, and a lone wave is bound by its own instruction stream (measured,
// scripts/ubench/lat.hip: a dependent fp64 add 4.3 cycles, a 2-dword DPP shift ~16, a ds_bpermute round trip 77-115
// cycles): the iteration of mg_gs_wave.h costs ~480 cycles -- two ds_bpermute round trips and, every iteration, a
// residual on 64 lanes with a 6-step reduction.  Here
//   * the 6 x 6 interior is held as 3 x 3 blocks of 2 x 2 points, one block per lane (lane = 4*block_row +
//     block_col, column 3 is a zero dummy), so that every neighbour is either in the lane's own registers or ONE
//     in-row DPP shift away (row_shr/shl:1 east-west, :4 north-south; dummy lanes and the lanes past the last block
//     row supply the rim zeros).  a, d are the red points of a block, b, c the black ones -- the same in every lane,
//     so there is no per-lane colour select either;
//   * the reference tests `err > target_error` after every sweep (:996), err = sum|r|/(N-2)^2 (:1051-1059).  The
//     red half of the NEXT sweep is computed first and held aside: its step a' - a IS the residual of that red point
//     in this iterate (times h^2/4, up to a rounding bounded in the code below), so one subtraction and one compare
//     per red point give a LOWER bound of the norm's sum.  While some lane's bound exceeds twice the target the
//     answer of the full test is known to be "go on" and the full norm -- four residuals and the reduction -- is
//     skipped; near convergence (the last ~10 iterations) the full norm runs every iteration.  The iterates, the
//     stopping iteration and the result are those of the reference.
// (A version that published every iterate to the other 15 waves for judging was measured slower than the lone-wave
// solver it replaced: every LDS operation costs the latency-critical wave 40-100 cycles.)
template <int CTRL>
__device__ __forceinline__ double gsp_shift(double v)  // in-row DPP shift, zero fill
{
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true),
                            __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true));
}

// wave 0 solves, the other waves wait at the closing barrier; N even, 4 <= N <= 8
__device__ __forceinline__ void gauss_seidel_blocks(int N, double h2, double inv, int src, int F, double tol, int *state)
{
    const int tid = threadIdx.x;
    if (tid < N * N) SRC(tid) = real_t(0.0);        // the rim (and, for now, everything else)
    if (tid < 16) {                                 // the 16 lanes of DPP row 0 carry the blocks
        const int lane = tid;
        const int nb = (N - 2) >> 1;                // blocks per side
        const int br = lane >> 2, bc = lane & 3;
        const bool active = br < nb && bc < nb;
        const int r0 = 1 + 2 * br, c0 = 1 + 2 * bc;
        const int p00 = r0 * N + c0;
        double fa = 0.0, fb = 0.0, fc = 0.0, fd = 0.0;
        if (active) {
            fa = (double)FF(p00);
            fb = (double)FF(p00 + 1);
            fc = (double)FF(p00 + N);
            fd = (double)FF(p00 + N + 1);
        }
        const double q = active ? 0.25 : 0.0;       // lanes without a block stay zero: they ARE the rim
        // 0.25*(s - h^2 F) as ONE fma, fma(0.25, s, -(0.25*h^2 F)): scaling by a power of two commutes with the
        // rounding (below the denormal range nothing here lives), so the bits are those of :1020's expression
        const double qa = q * (h2 * fa), qb = q * (h2 * fb), qc = q * (h2 * fc), qd = q * (h2 * fd);
        const double keep = active ? 1.0 : 0.0;
        const double denom = (double)((N - 2) * (N - 2));
        // `sum/denom > tol` without the division wherever rounding cannot matter (see gsw::solve)
        const double thr = tol * denom, thr_hi = thr * (1.0 + 0x1p-48), thr_lo = thr * (1.0 - 0x1p-48);
        // Judging an iterate without the norm.  The red update of the NEXT sweep, a' = 0.25*(S - h^2 F), is built from
        // the very neighbour sum S the residual of that red point in THIS iterate is, r = inv*(S - 4a) - F:
        // r = 4*inv*(a' - a) up to rounding, E <= eps*(60*inv*max|U| + 6*|F|), and the black residuals of an iterate
        // (black was updated last) are rounding alone.  Zero start and |u_new| <= max|neighbours| + h^2|F|/4 give
        // max|U| <= 2k*h^2*max|F|/4 after k sweeps, so E <= 64*k*eps*max|F| per point.  Hence
        //   (1) a lane with 4*inv*|a' - a| > 2*thr has |r| > 1.5*thr: the norm's sum, all terms non-negative, is above;
        //   (2) est = sum over the lanes of |a' - a| + |d' - d| is the norm's sum / (4*inv) to within 40*E/(4*inv): outside
        //       a band of 2^-10 around the target it decides, inside the band the norm itself is computed;
        //   (3) a half-sweep zeroes the residuals of its colour and moves a quarter of each to the (at most four)
        //       neighbours of the other colour, so the L1 norm of the residual never grows from one iterate to the
        //       next: an iterate followed, any number of sweeps later, by one that is certainly above the target was
        //       above it too.
        // All three hold while 40*64*k*eps*max|F| <= thr*2^-10, i.e. k <= thr*2^31/max|F| (7700 sweeps at the shipped
        // tolerance and |F| = 1; the coarse right-hand side of a cycle is a residual, far smaller); past that every
        // iterate gets the norm.  The solve therefore runs GS_BATCH sweeps between tests -- for the lone wave a test
        // and its taken branch cost as much as a sweep and a half (scripts/ubench/gs_iter.hip) -- remembers the red
        // values the batch started from (red-black: they ARE the state), and when a batch ends at or below the target
        // goes back to them and repeats those sweeps judging each iterate: the stopping iterate, its bits and its
        // number are the reference's.
        constexpr int GS_BATCH = 8;
        double fm = fmax(fmax(fabs(fa), fabs(fb)), fmax(fabs(fc), fabs(fd)));
        fm = fmax(fm, gsw::row_shr_zero<1>(fm));
        fm = fmax(fm, gsw::row_shr_zero<2>(fm));
        fm = fmax(fm, gsw::row_shr_zero<4>(fm));
        fm = fmax(fm, gsw::row_shr_zero<8>(fm));
        const double f_max = gsw::read_lane(fm, 15), k_num = thr * 0x1p31;
        int fast_until = 50000000;                 // the usual case, without the division; F = 0 lands here too
        if (!(k_num >= 5e7 * f_max)) {             // (a NaN anywhere lands in this branch, and at 0)
            const double k_safe = k_num / f_max;
            fast_until = __builtin_amdgcn_readfirstlane(k_safe > 0.0 ? (int)k_safe : 0);
        }
        // h2 for 1/inv: the tests carry a margin of 2^-10 and more, the last bits of their thresholds are free
        const double thr_e = 0.25 * thr * h2, thr_d = 2.0 * thr_e, thr_e_hi = thr_e * (1.0 + 0x1p-10), thr_e_lo = thr_e * (1.0 - 0x1p-10);
        struct Iterate {
            double a, b, c, d;              // the block: a, d red; b, c black
            double a_e, d_s, d_w, a_n;      // lane+1's a, lane+4's d, lane-1's d, lane-4's a
            double b_w, c_s, c_e, b_n;      // lane-1's b, lane-4's c, lane+1's c, lane+4's b
        };
        // one sweep: the red values (na, nd) prepared by the sweep before, the black half, and the red half of the next
        // sweep kept aside until this iterate has been judged (:996)
        auto sweep = [&](Iterate &s, double &na, double &nd) {
            s.a = na;                       // red :1020, U = 0.25*(U[l] + U[r] + U[t] + U[b] - h^2 F)
            s.d = nd;
            s.a_e = gsp_shift<0x101>(s.a), s.d_s = gsp_shift<0x114>(s.d), s.d_w = gsp_shift<0x111>(s.d), s.a_n = gsp_shift<0x104>(s.a);
            s.b = __builtin_fma(q, ((s.a + s.a_e) + s.d) + s.d_s, -qb);   // black :1043
            s.c = __builtin_fma(q, ((s.d_w + s.d) + s.a_n) + s.a, -qc);
            s.b_w = gsp_shift<0x111>(s.b);
            s.c_s = gsp_shift<0x114>(s.c);
            s.c_e = gsp_shift<0x101>(s.c);
            s.b_n = gsp_shift<0x104>(s.b);
            na = __builtin_fma(q, ((s.b_w + s.b) + s.c) + s.c_s, -qa);    // west + east + north + south, as mg_gs_wave.h
            nd = __builtin_fma(q, ((s.c + s.c_e) + s.b_n) + s.b, -qd);
        };
        auto row_total = [](double acc) {
            acc += gsw::row_shr_zero<1>(acc);
            acc += gsw::row_shr_zero<2>(acc);
            acc += gsw::row_shr_zero<4>(acc);
            acc += gsw::row_shr_zero<8>(acc);       // lane 15 holds the total
            return gsw::read_lane(acc, 15);
        };
        // the norm itself (:560, :1051-1059): inv*(n + s + e + w - 4u) - f over the four points of the block
        auto above_target = [&](const Iterate &s) {
            const double ra = inv * (s.c + s.c_s + s.b + s.b_w - 4 * s.a) - fa;
            const double rd = inv * (s.b_n + s.b + s.c_e + s.c - 4 * s.d) - fd;
            const double rb = inv * (s.d + s.d_s + s.a_e + s.a - 4 * s.b) - fb;
            const double rc = inv * (s.a_n + s.a + s.d + s.d_w - 4 * s.c) - fc;
            const double sum = row_total(keep * (fabs(ra) + fabs(rb) + fabs(rc) + fabs(rd)));
            return sum > thr_hi ? true : (sum < thr_lo ? false : sum / denom > tol);  // :1059, :996
        };
        Iterate u;
        double na = 0.0 - qa, nd = 0.0 - qd;   // the first red half-sweep, from memset(U, 0) :993  (0 - 0 = +0, as 0.25*(0 - h^2*0))
        int iterations = 0;
        while (iterations + GS_BATCH <= fast_until) {
            const double batch_a = na, batch_d = nd;
#pragma unroll
            for (int m = 0; m < GS_BATCH; ++m) sweep(u, na, nd);
            const double step_a = fabs(na - u.a), step_d = fabs(nd - u.d);
            bool above = __builtin_amdgcn_ballot_w64(fmax(step_a, step_d) > thr_d) != 0;              // (1)
            if (!above) above = row_total(keep * (step_a + step_d)) > thr_e_hi;                       // (2)
            if (!above) {                   // the stop lies in this batch: once more, judging every iterate
                na = batch_a;
                nd = batch_d;
                break;
            }
            iterations += GS_BATCH;         // (3): and so were the iterates in between
        }
        for (;;) {
            sweep(u, na, nd);
            ++iterations;
            bool above, decided = false;
            if (iterations < fast_until) {
                const double est = row_total(keep * (fabs(na - u.a) + fabs(nd - u.d)));               // (2)
                above = est > thr_e_hi;
                decided = above || est < thr_e_lo;
            }
            if (!decided) above = above_target(u);
            if (!above || iterations >= 50000000) break;
        }
        if (active) {
            SRC(p00) = (real_t)u.a;
            SRC(p00 + 1) = (real_t)u.b;
            SRC(p00 + N) = (real_t)u.c;
            SRC(p00 + N + 1) = (real_t)u.d;
        }
        if (lane == 0) {
            state[0] = 1;
            state[1] = iterations;
        }
    }
    __syncthreads();
}

#define ITAB(i) (reinterpret_cast<int *>(lds)[(i)])

// first double past the level arrays (fp32 fields only: fp64 scratch of the exact solver)
__host__ __device__ __forceinline__ int gs_scratch(const TailArgsT<real_t> &a)
{
    size_t elems = 0;
    for (int l = 0; l < a.n_levels; ++l) elems += (size_t)3 * a.N[l] * a.N[l];
    return (int)((elems * sizeof(real_t) + 7) / 8);
}

// diagnostics (-DMG_TAIL_PHASES, with MG_TAIL_TRACE=1): shader-clock stamps of thread 0 inside the first 24 nodes
#ifdef MG_TAIL_PHASES
#define PHASE(k) do { if (a.trace && threadIdx.x == 0 && i < 24) a.trace[TAIL_MAX_NODES + 2 + i * 8 + (k)] = clock64(); } while (0)
#else
#define PHASE(k) do { } while (0)
#endif

// lane l's value of a per-lane table, for a wave-uniform l
__device__ __forceinline__ int lane_get(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float lane_get(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ double lane_get(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

__global__ __launch_bounds__(TAIL_THREADS) void k_tail(const TailArgsT<real_t> a)
{
    __shared__ double sm[17];                 // block_total of the block Gauss-Seidel
    __shared__ double slots[2][TAIL_WAVES];   // error partials, alternating between nodes
    // The node program and the per-level constants live in the LANES of a few registers of every wave (lane i: node i,
    // lane l: level l) and are fetched with v_readlane.  Read from the kernel arguments with a run-time index they are
    // scalar loads, and a lone dependent scalar load costs a wave ~250 cycles: several per node, and one in every
    // sweep for the offset of the level's arrays, were a third of a node's time on the 16 x 16 level.
    // the arrays of this instance: a batch (the independent coarse-tail slices of a W-cycle, one workgroup each) reads them
    // from a table in device memory
    const real_t *a_F_top = a.F_top;
    real_t *a_U_top = a.U_top;
    double *a_err_dev = a.err_dev;
    int *a_gs_state = a.gs_state;
    if (a.batch) {
        typedef const TailBatchItem __attribute__((address_space(4))) *item_ptr;
        const item_ptr b = (item_ptr)(uintptr_t)(a.batch + blockIdx.x);
        a_F_top = static_cast<const real_t *>(b->F_top);
        a_U_top = static_cast<real_t *>(b->U_top);
        a_err_dev = b->err_dev;
        a_gs_state = b->gs_state;
    }
    const int tab_lane = threadIdx.x & 63;
    int lv_where = 0, lv_tabs = 0;            // base | N << 16;  real tables | int tables << 16  (offsets < 2^16: 160 KB of LDS)
    real_t lv_dx2 = 0, lv_inv = 0, lv_cdx = 1, lv_crcp = 1;
    double lv_gs_h2 = 0, lv_gs_inv = 0;
    if (tab_lane < a.n_levels) {
        int base = 0, rt = a.tab_real0, it = a.tab_int0;
#pragma unroll
        for (int k = 0; k < TAIL_MAX_LEVELS - 1; ++k)
            if (k < tab_lane) {
                base += 3 * a.N[k] * a.N[k];
                rt += a.N[k + 1] + 4 * a.N[k];   // reals  w[M] rhi[N] rlo[N] chi[N] clo[N]  of level pair k <-> k+1
                it += a.N[k + 1] + 2 * a.N[k];   // ints   lo[M] orow[N] ocol[N]
            }
        lv_where = base | (a.N[tab_lane] << 16);
        lv_tabs = rt | (it << 16);
        lv_dx2 = a.dx2[tab_lane];
        lv_inv = a.inv[tab_lane];
        lv_gs_h2 = a.gs_h2[tab_lane];
        lv_gs_inv = a.gs_inv[tab_lane];
        if (tab_lane + 1 < a.n_levels) {
            lv_cdx = a.c_dx[tab_lane];
            lv_crcp = real_t(1.0) / lv_cdx;
        }
    }
    int nd_what = 1, nd_err = -1;             // type + 1 | steps << 2
    double nd_tol = 0.0;
    if (tab_lane < a.n_nodes) {
        nd_what = (a.nodes[tab_lane].type + 1) | (a.nodes[tab_lane].steps << 2);
        nd_err = a.nodes[tab_lane].err_slot;
        nd_tol = a.nodes[tab_lane].tol;
    }
    auto N_of = [&](int l) { return lane_get(lv_where, l) >> 16; };
    auto real_tab_of = [&](int l) { return lane_get(lv_tabs, l) & 0xffff; };
    auto int_tab_of = [&](int l) { return (int)((unsigned)lane_get(lv_tabs, l) >> 16); };
    unsigned swapped = 0;  // bit l: level l's U currently lives in its second buffer (same in every thread)
    auto U_of = [&](int l) { const int w = lane_get(lv_where, l), n = w >> 16; return (w & 0xffff) + (((swapped >> l) & 1u) ? n * n : 0); };
    auto T_of = [&](int l) { const int w = lane_get(lv_where, l), n = w >> 16; return (w & 0xffff) + (((swapped >> l) & 1u) ? 0 : n * n); };
    auto F_of = [&](int l) { const int w = lane_get(lv_where, l), n = w >> 16; return (w & 0xffff) + 2 * n * n; };
    if (a.trace && threadIdx.x == 0) a.trace[0] = wall_clock64();
    {
        // stage the finest source and all transfer tables: one global round trip for the launch
        const int N0 = N_of(0), f0 = F_of(0);
        for (int p = threadIdx.x; p < N0 * N0; p += TAIL_THREADS) lds[f0 + p] = a_F_top[p];
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        for (int l = 0; l + 1 < a.n_levels; ++l) {
            const int N = N_of(l), M = N_of(l + 1), rt = real_tab_of(l), it = int_tab_of(l);
            constexpr int W = TAIL_WAVES < 8 ? TAIL_WAVES : 8;   // the eight tables over the first W waves
            for (int i = lane; i < N; i += 64) {
                if (wave == 0 % W && i < M) lds[rt + i] = a.r_w[l][i];
                if (wave == 1 % W) lds[rt + M + i] = a.p_rhi[l][i];
                if (wave == 2 % W) lds[rt + M + N + i] = a.p_rlo[l][i];
                if (wave == 3 % W) lds[rt + M + 2 * N + i] = a.p_chi[l][i];
                if (wave == 4 % W) lds[rt + M + 3 * N + i] = a.p_clo[l][i];
                if (wave == 5 % W && i < M) ITAB(it + i) = a.r_lo[l][i];
                if (wave == 6 % W) ITAB(it + M + i) = a.p_orow[l][i];
                if (wave == 7 % W) ITAB(it + M + N + i) = a.p_ocol[l][i];
            }
        }
    }
    __syncthreads();
    if (a.trace && threadIdx.x == 0) a.trace[1] = wall_clock64();

    int cur = 0, parity = 0;
    // The error of a node's smoothing (:621-622: sixteen partial sums, two divisions, a store) is finished by the LAST
    // wave at the start of the next node: on the small levels and during the exact solve that wave owns no points,
    // and the 600 cycles leave the critical path.
    int pend_slot = -1, pend_N = 0, pend_parity = 0;
    auto flush_error = [&]() {
        if (pend_slot >= 0 && threadIdx.x == TAIL_THREADS - 64) finish_error(slots[pend_parity], pend_N, a_err_dev + pend_slot);
        pend_slot = -1;
    };
    const int n_nodes = a.n_nodes;
    for (int i = 0; i < n_nodes; ++i) {
        TailNode nd;
        {
            const int w = lane_get(nd_what, i);
            nd.type = (w & 3) - 1;
            nd.steps = w >> 2;
            nd.err_slot = lane_get(nd_err, i);
            nd.tol = lane_get(nd_tol, i);
        }
        if (a.trace && threadIdx.x == 0 && i > 0) a.trace[1 + i] = wall_clock64();
        flush_error();
        // A wave none of whose threads owns a point of this node's level (N = 16: 12 of the 16 waves) only
        // keeps the barrier count: the four waves of a SIMD issue in turn, so every instruction an idle wave
        // runs through delays the wave next to it that has the work.
        // (readfirstlane: the compiler must see a wave-uniform value, or every variable updated under the
        // branch -- cur, swapped, parity -- turns into a per-lane register and the loops below into masked ones)
        const int wave_first = __builtin_amdgcn_readfirstlane((int)threadIdx.x) & ~63;
        if (nd.type == -1) {
            // memset(U,0) :256, doSmoothing :259, getResidual :268, sign flip :277-280, doRestriction :287
            const int N = N_of(cur);
            if (wave_first >= N * N) {
                if ((threadIdx.x & 63) == 0) slots[parity][threadIdx.x >> 6] = 0.0;
                for (int b = 0; b < nd.steps + 2; ++b) __syncthreads();
                if ((nd.steps - 1) & 1) swapped ^= 1u << cur;
                pend_slot = nd.err_slot, pend_N = N, pend_parity = parity;
                parity ^= 1;
                ++cur;
                continue;
            }
            auto down = [&](auto np_tag) {
            constexpr int NP = decltype(np_tag)::value;
            PHASE(0);
            const int F = F_of(cur);
            const PointsT<NP> P = map_points<NP>(N);
            const real_t dx2 = lane_get(lv_dx2, cur), inv = lane_get(lv_inv, cur);
            real_t v[NP], f[NP], h2f[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                f[k] = P.live[k] ? lds[F + P.p[k]] : real_t(0.0);
                h2f[k] = dx2 * f[k];
                v[k] = 0.0;
            }
            PHASE(1);
            sweep<true, NP>(P, N, 0, U_of(cur), v, h2f);
            PHASE(2);
            for (int s = 1; s < nd.steps; ++s) {
                sweep<false, NP>(P, N, U_of(cur), T_of(cur), v, h2f);
                swapped ^= 1u << cur;
            }
            PHASE(3);
            // residual once per point: the error norm (:607-622) and, negated, the restriction's input
            const int src = U_of(cur), D = T_of(cur);
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                if (!P.live[k]) continue;
                const int p = P.p[k];
                real_t d = 0.0;
                if (P.inner[k]) {
                    d = inv * (SRC(p + N) + SRC(p - N) + SRC(p + 1) + SRC(p - 1) - 4 * v[k]) - f[k];
                    if (P.even[k]) acc += fabs((double)d);
                }
                lds[D + p] = -d;
            }
            post_partial(acc, slots[parity]);
            __syncthreads();
            PHASE(4);
            pend_slot = nd.err_slot, pend_N = N, pend_parity = parity;
            PHASE(5);
            parity ^= 1;
            const int M = N_of(cur + 1), Fc = F_of(cur + 1);
            const int rt = real_tab_of(cur), it = int_tab_of(cur);
            constexpr int NQ = (TAIL_MAX_N / 2) * (TAIL_MAX_N / 2) <= TAIL_THREADS ? 1 : (NP + 3) / 4;  // coarse points per thread
            const PointsT<NQ> Q = map_points<NQ>(M);
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                if (!Q.live[k]) continue;
                real_t vc = 0.0;
                if (Q.inner[k]) {
                    const int rc = Q.r[k], cc = Q.c[k];
                    const real_t wa = lds[rt + cc], wb = real_t(1.0) - wa, wc = lds[rt + rc], wd = real_t(1.0) - wc;
                    const int q = D + ITAB(it + rc) * N + ITAB(it + cc);
                    const real_t u0 = lds[q], u1 = lds[q + 1], u2 = lds[q + N], u3 = lds[q + N + 1];
                    vc = wb * wd * u0 + wa * wd * u1 + wc * wb * u2 + wa * wc * u3;  // :676
                }
                lds[Fc + Q.p[k]] = vc;
            }
            };
            if (N * N <= TAIL_THREADS) down(std::integral_constant<int, 1>{});
            else if (N * N <= 2 * TAIL_THREADS) down(std::integral_constant<int, 2>{});
            else if (PT > 4 && N * N <= 4 * TAIL_THREADS) down(std::integral_constant<int, 4>{});
            else down(std::integral_constant<int, PT>{});
            __syncthreads();
            PHASE(6);
            ++cur;
        } else if (nd.type == 0) {
            const int N = N_of(cur);
            const double gs_h2 = lane_get(lv_gs_h2, cur), gs_inv = lane_get(lv_gs_inv, cur);
            if (N * N <= 64 && (N & 1) == 0 && N >= 4) {
                gauss_seidel_blocks(N, gs_h2, gs_inv, U_of(cur), F_of(cur), nd.tol, a_gs_state);
            } else if (N * N <= 64) {
                gauss_seidel_wave(N, gs_h2, gs_inv, U_of(cur), F_of(cur), nd.tol, a_gs_state);
            } else if (N * N <= 64 * GS_WAVE_PTS) {
                // fp64 fields: the level's own U array is the solver's array; fp32 fields: the fp64 scratch
                const bool in_place = sizeof(real_t) == sizeof(double);
                gauss_seidel_wave_lds(N, gs_h2, gs_inv, in_place ? U_of(cur) : gs_scratch(a), in_place, U_of(cur), F_of(cur),
                                      nd.tol, a_gs_state);
            } else gauss_seidel_level(N, gs_h2, gs_inv, U_of(cur), F_of(cur), gs_scratch(a), nd.tol, sm, a_gs_state);
            __syncthreads();
        } else {  // 1: doProlongation :354, doGridAddition :368, doSmoothing :416
            const int Nc = N_of(cur), fine = cur - 1, N = N_of(fine);
            if (wave_first >= N * N) {
                if ((threadIdx.x & 63) == 0) slots[parity][threadIdx.x >> 6] = 0.0;
                for (int b = 0; b < nd.steps + 2; ++b) __syncthreads();
                if (nd.steps & 1) swapped ^= 1u << fine;
                pend_slot = nd.err_slot, pend_N = N, pend_parity = parity;
                parity ^= 1;
                --cur;
                continue;
            }
            auto up = [&](auto np_tag) {
            constexpr int NP = decltype(np_tag)::value;
            PHASE(0);
            const int uc = U_of(cur), F = F_of(fine);
            const int rt = real_tab_of(fine) + Nc, it = int_tab_of(fine) + Nc;  // past w[M] / lo[M]
            const real_t c_dx = lane_get(lv_cdx, fine), c_rcp = lane_get(lv_crcp, fine), dx2 = lane_get(lv_dx2, fine), inv = lane_get(lv_inv, fine);
            const PointsT<NP> P = map_points<NP>(N);
            real_t v[NP], f[NP], h2f[NP];
            {
                const int uf = U_of(fine);
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    f[k] = real_t(0.0);
                    v[k] = real_t(0.0);
                    if (P.live[k]) {
                        const int q = P.p[k], kf = P.r[k], l = P.c[k];
                        f[k] = lds[F + q];
                        v[k] = lds[uf + q];
                        const int ci = ITAB(it + kf), cj = ITAB(it + N + l);
                        if (ci >= 0 && cj >= 0) {
                            const int p = uc + ci * Nc + cj;
                            const real_t c1 = lds[p], c2 = lds[p + 1], c3 = lds[p + Nc], c4 = lds[p + Nc + 1];
                            const real_t rhi = lds[rt + kf], rlo = lds[rt + N + kf], chi = lds[rt + 2 * N + l], clo = lds[rt + 3 * N + l];
                            const real_t num = (c1 * chi + c2 * clo) * rhi + (c3 * chi + c4 * clo) * rlo;
                            v[k] = v[k] + div_by_const(div_by_const(num, c_dx, c_rcp), c_dx, c_rcp);  // :700 .../c_dx/c_dx
                            lds[uf + q] = v[k];
                        }
                    }
                    h2f[k] = dx2 * f[k];
                }
            }
            __syncthreads();
            PHASE(1);
            for (int s = 0; s < nd.steps; ++s) {
                sweep<false, NP>(P, N, U_of(fine), T_of(fine), v, h2f);
                swapped ^= 1u << fine;
            }
            PHASE(2);
            const int src = U_of(fine);
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                if (P.inner[k] && P.even[k]) {
                    const int p = P.p[k];
                    acc += fabs((double)(inv * (SRC(p + N) + SRC(p - N) + SRC(p + 1) + SRC(p - 1) - 4 * v[k]) - f[k]));
                }
            }
            post_partial(acc, slots[parity]);
            __syncthreads();
            PHASE(3);
            pend_slot = nd.err_slot, pend_N = N, pend_parity = parity;
            PHASE(4);
            };
            if (N * N <= TAIL_THREADS) up(std::integral_constant<int, 1>{});
            else if (N * N <= 2 * TAIL_THREADS) up(std::integral_constant<int, 2>{});
            else if (PT > 4 && N * N <= 4 * TAIL_THREADS) up(std::integral_constant<int, 4>{});
            else up(std::integral_constant<int, PT>{});
            parity ^= 1;
            --cur;
        }
    }
    flush_error();
    if (a.trace && threadIdx.x == 0) a.trace[1 + a.n_nodes] = wall_clock64();
    {
        const int N0 = N_of(0), u0 = U_of(0);
        for (int p = threadIdx.x; p < N0 * N0; p += TAIL_THREADS) a_U_top[p] = lds[u0 + p];
    }
}

// LDS layout: level arrays | (fp32 fields) fp64 scratch of the block exact solver | real tables | int tables
struct TailLayout {
    int tab_real0, tab_int0;
    size_t bytes;
};
inline TailLayout tail_layout(const TailArgsT<real_t> &a)
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
    bytes = (bytes + 7) / 8 * 8;
    size_t n_real = 0, n_int = 0;
    for (int l = 0; l + 1 < a.n_levels; ++l) {
        n_real += (size_t)a.N[l + 1] + 4 * (size_t)a.N[l];
        n_int += (size_t)a.N[l + 1] + 2 * (size_t)a.N[l];
    }
    TailLayout L;
    L.tab_real0 = (int)(bytes / sizeof(real_t));
    bytes += (n_real * sizeof(real_t) + 7) / 8 * 8;
    L.tab_int0 = (int)(bytes / sizeof(int));
    bytes += n_int * sizeof(int);
    L.bytes = bytes;
    return L;
}
inline size_t tail_lds_bytes(const TailArgsT<real_t> &a) { return tail_layout(a).bytes; }

inline void tail_launch(hipStream_t s, const TailArgsT<real_t> &a, int n_batch = 1, const TailBatchItem *batch_dev = nullptr)
{
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)k_tail, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
        attr_set = true;
    }
    const TailLayout L = tail_layout(a);
    TailArgsT<real_t> b = a;
    b.tab_real0 = L.tab_real0;
    b.tab_int0 = L.tab_int0;
    b.batch = batch_dev;   // one workgroup (one CU: the levels fill its LDS) per instance
    hipLaunchKernelGGL(k_tail, dim3(batch_dev ? n_batch : 1), dim3(TAIL_THREADS), L.bytes, s, b);
}

}  // namespace MG_REAL_NS
}  // namespace k
}  // namespace mg
