// lat.hip -- single-wave instruction latencies on gfx950 that bound the one-wave Gauss-Seidel solve
// (mg_gs_wave.h): dependent fp64 add/mul chains, DPP moves, ds_bpermute, v_permlane*_swap, LDS round trips,
// s_barrier with 16 waves.  Prints cycles per operation (s_memtime shader clock).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/lat.hip -o scripts/ubench/lat.bin && scripts/ubench/lat.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 256
// the clock read must not move across the measured chain: tie it to the chain's value through inline asm
__device__ __forceinline__ long long now(double &x)
{
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(x)::"memory");
    long long t = __builtin_readcyclecounter();
    asm volatile("s_nop 0" : "+v"(x), "+s"(t)::"memory");
    return t;
}

__device__ __forceinline__ double dpp_shr1(double v)
{
    union { double d; int i[2]; } a, r;
    a.d = v;
    r.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], 0x138, 0xf, 0xf, true);
    r.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], 0x138, 0xf, 0xf, true);
    return r.d;
}
__device__ __forceinline__ double dpp_ror8(double v)
{
    union { double d; int i[2]; } a, r;
    a.d = v;
    r.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], 0x128, 0xf, 0xf, true);
    r.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], 0x128, 0xf, 0xf, true);
    return r.d;
}

__global__ void k_lat(double *out, long long *t, double seed, int lane_n)
{
    __shared__ double sh[1024];
    const int lane = threadIdx.x & 63;
    double x = seed + lane * 1e-9, y = 1.0 + seed;
    sh[threadIdx.x] = x;
    __syncthreads();
    long long t0, t1;
    int slot = 0;
    // (0) dependent v_add_f64
    t0 = now(x);
#pragma unroll
    for (int i = 0; i < REP; ++i) x = x + y;
    t1 = now(x);
    if (threadIdx.x == 0) t[slot] = t1 - t0;
    ++slot;
    // (1) dependent v_mul_f64
    t0 = now(x);
#pragma unroll
    for (int i = 0; i < REP; ++i) x = x * y;
    t1 = now(x);
    if (threadIdx.x == 0) t[slot] = t1 - t0;
    ++slot;
    // (2) dependent DPP shift (2 movs) + add
    t0 = now(x);
#pragma unroll
    for (int i = 0; i < REP; ++i) x = x + dpp_shr1(x);
    t1 = now(x);
    if (threadIdx.x == 0) t[slot] = t1 - t0;
    ++slot;
    // (3) dependent ds_bpermute (2 dwords) + add
    t0 = now(x);
#pragma unroll
    for (int i = 0; i < REP; ++i) x = x + __shfl(x, lane_n, 64);
    t1 = now(x);
    if (threadIdx.x == 0) t[slot] = t1 - t0;
    ++slot;
    // (4) dependent permlane16_swap pair + add
    t0 = now(x);
#pragma unroll
    for (int i = 0; i < REP; ++i) {
        union { double d; int i[2]; } a, b;
        a.d = x;
        b.d = x;
        auto r0 = __builtin_amdgcn_permlane16_swap(a.i[0], b.i[0], false, false);
        auto r1 = __builtin_amdgcn_permlane16_swap(a.i[1], b.i[1], false, false);
        a.i[0] = r0[1];
        a.i[1] = r1[1];
        x = x + a.d;
    }
    t1 = now(x);
    if (threadIdx.x == 0) t[slot] = t1 - t0;
    ++slot;
    // (5) dependent LDS write + read (own wave's slots, other lane)
    t0 = now(x);
#pragma unroll
    for (int i = 0; i < REP; ++i) {
        sh[threadIdx.x] = x;
        x = x + sh[(threadIdx.x & ~63) + lane_n];
    }
    t1 = now(x);
    if (threadIdx.x == 0) t[slot] = t1 - t0;
    ++slot;
    // (6) s_barrier with all waves of the block
    t0 = now(x);
#pragma unroll 16
    for (int i = 0; i < REP; ++i) __syncthreads();
    t1 = now(x);
    if (threadIdx.x == 0) t[slot] = t1 - t0;
    ++slot;
    // (7) independent v_add_f64 stream (issue rate)
    double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
    t0 = now(x);
#pragma unroll
    for (int i = 0; i < REP / 8; ++i) {
        a0 += y; a1 += y; a2 += y; a3 += y; a4 += y; a5 += y; a6 += y; a7 += y;
    }
    x = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    t1 = now(x);
    if (threadIdx.x == 0) t[slot] = t1 - t0;
    ++slot;
    // (8) ror8 DPP + add dependent
    t0 = now(x);
#pragma unroll
    for (int i = 0; i < REP; ++i) x = x + dpp_ror8(x);
    t1 = now(x);
    if (threadIdx.x == 0) t[slot] = t1 - t0;
    ++slot;
    out[threadIdx.x] = x + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main()
{
    double *out;
    long long *t;
    hipMalloc(&out, 1024 * sizeof(double));
    hipMalloc(&t, 16 * sizeof(long long));
    const char *names[] = {"dependent v_add_f64", "dependent v_mul_f64", "DPP wave_shr:1 (2 movs) + add", "ds_bpermute x2 + add",
                           "permlane16_swap x2 + add", "LDS write + read + add", "s_barrier", "independent v_add_f64 (issue)", "DPP row_ror:8 + add"};
    for (int threads : {64, 1024}) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_lat, dim3(1), dim3(threads), 0, 0, out, t, 1e-3, 8);
        hipDeviceSynchronize();
        std::vector<long long> h(16);
        hipMemcpy(h.data(), t, 16 * sizeof(long long), hipMemcpyDeviceToHost);
        printf("block of %d threads (wave 0 timed), shader-clock cycles per operation:\n", threads);
        for (int i = 0; i < 9; ++i) printf("  %-34s %7.1f\n", names[i], (double)h[i] / REP);
    }
    return 0;
}
