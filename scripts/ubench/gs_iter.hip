// gs_iter.hip -- what one iteration of the block Gauss-Seidel solve (mg_tail_impl.h: gauss_seidel_blocks) costs a
// lone wave, and why: the same sweep (2 x 2 blocks per lane, in-row DPP shifts) run a fixed number of times, timed
// with the 100 MHz wall clock AND the shader clock counter (their ratio is the clock the wave really ran at), in
// variants that take one ingredient away at a time (the DPP shifts; the fp64 arithmetic; the other 15 waves of the
// workgroup parked at a barrier).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off scripts/ubench/gs_iter.hip -o scripts/ubench/gs_iter.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CTRL>
__device__ __forceinline__ double shift(double v)
{
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true),
                            __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true));
}

// M sweeps between two tests (the loop body is M sweeps, one |a' - a| test, one branch); DPP = false replaces the
// shifts by register copies, WIDE = false runs the same sweep on 32-bit floats (one DPP mov per shift)
template <int M, bool DPP, typename T>
__global__ void k_gs(double *out, long long *t, int iters, double seed)
{
    if (threadIdx.x < 16) {
        const int lane = threadIdx.x;
        const bool active = (lane >> 2) < 3 && (lane & 3) < 3;
        const T q = active ? T(0.25) : T(0);
        const T qa = q * T(seed * (1 + lane)), qb = q * T(seed * (2 + lane)), qc = q * T(seed * (3 + lane)), qd = q * T(seed * (4 + lane));
        T a = 0, b = 0, c = 0, d = 0, na = -qa, nd = -qd;
        const T thr_d = T(-1);   // never met: the loop runs its full count
        auto sh = [](T v, auto ctrl) {
            constexpr int C = decltype(ctrl)::value;
            if constexpr (!DPP) return v;
            else if constexpr (sizeof(T) == 8) return (T)shift<C>(v);
            else return (T)__int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), C, 0xf, 0xf, true));
        };
        auto fma_ = [](T x, T y, T z) { if constexpr (sizeof(T) == 8) return (T)__builtin_fma(x, y, z); else return (T)__builtin_fmaf(x, y, z); };
        using E = std::integral_constant<int, 0x101>;
        using S = std::integral_constant<int, 0x114>;
        using W = std::integral_constant<int, 0x111>;
        using Nn = std::integral_constant<int, 0x104>;
        long long w0 = wall_clock64(), c0 = clock64();
        int n = 0;
        for (;;) {
#pragma unroll
            for (int m = 0; m < M; ++m) {
                a = na;
                d = nd;
                const T a_e = sh(a, E()), d_s = sh(d, S()), d_w = sh(d, W()), a_n = sh(a, Nn());
                b = fma_(q, ((a + a_e) + d) + d_s, -qb);
                c = fma_(q, ((d_w + d) + a_n) + a, -qc);
                const T b_w = sh(b, W()), c_s = sh(c, S()), c_e = sh(c, E()), b_n = sh(b, Nn());
                na = fma_(q, ((b_w + b) + c) + c_s, -qa);
                nd = fma_(q, ((c + c_e) + b_n) + b, -qd);
            }
            n += M;
            const T da = na - a, dd = nd - d;
            const T mx = (da < 0 ? -da : da) > (dd < 0 ? -dd : dd) ? (da < 0 ? -da : da) : (dd < 0 ? -dd : dd);
            if (n < iters && __builtin_amdgcn_ballot_w64(mx > thr_d) != 0) continue;
            break;
        }
        long long w1 = wall_clock64(), c1 = clock64();
        if (lane == 0) {
            t[0] = w1 - w0;
            t[1] = c1 - c0;
            t[2] = n;
        }
        out[lane] = (double)(na + nd + b + c);
    }
    __syncthreads();
}

template <int M, bool DPP, typename T>
static void run(const char *name, int threads, double *out, long long *t)
{
    const int iters = 16000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_gs<M, DPP, T>), dim3(1), dim3(threads), 0, 0, out, t, iters, 1e-3);
    (void)hipDeviceSynchronize();
    long long h[3];
    (void)hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    const double ns = h[0] * 10.0 / h[2], cyc = (double)h[1] / h[2];
    printf("%-44s %4d threads: %6.1f ns = %6.1f shader cycles per iteration (clock %.0f MHz)\n", name, threads, ns, cyc, cyc / ns * 1e3);
}

int main()
{
    double *out;
    long long *t;
    (void)hipMalloc(&out, 64 * sizeof(double));
    (void)hipMalloc(&t, 4 * sizeof(long long));
    for (int threads : {1024, 64}) {
        run<1, true, double>("fp64, test after every sweep", threads, out, t);
        run<2, true, double>("fp64, test after every 2nd sweep", threads, out, t);
        run<4, true, double>("fp64, test after every 4th sweep", threads, out, t);
        run<8, true, double>("fp64, test after every 8th sweep", threads, out, t);
        run<16, true, double>("fp64, test after every 16th sweep", threads, out, t);
        run<8, false, double>("fp64, 8 sweeps per test, no DPP", threads, out, t);
        run<8, true, float>("fp32, 8 sweeps per test", threads, out, t);
        run<8, false, float>("fp32, 8 sweeps per test, no DPP", threads, out, t);
    }
    return 0;
}
