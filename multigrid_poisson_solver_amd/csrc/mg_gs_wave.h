// mg_gs_wave.h -- red-black Gauss-Seidel (src/MG_solver_CPU.cpp:952-1066) for a grid of at most
// 64 points by ONE wave: one point per lane, U in a register.  Shared by the stand-alone solver
// kernel (mg_kernels.hip:k_gs_wave) and the coarse-tail kernel (mg_tail.hip).
//
// East/west neighbours are the adjacent lanes (a full-wave DPP shift), north/south are N lanes
// away (ds_bpermute).  The residual norm needs the neighbours of the state after the black
// pass -- exactly what the red pass of the NEXT iteration needs, so each iteration fetches
// neighbours twice, not three times.  The norm is a DPP row scan + 4 readlanes; convergence is
// tested every iteration like the reference's `while (err > target_error)` (:996).
#pragma once
#include <hip/hip_runtime.h>

namespace mg {
namespace k {
namespace gsw {

__device__ __forceinline__ double lane_shift(double v, int dpp_ctrl_is_shr)
{
    union { double d; int i[2]; } a, r;
    a.d = v;
    if (dpp_ctrl_is_shr) {
        r.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
        r.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], 0x138, 0xf, 0xf, true);
    } else {
        r.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
        r.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], 0x130, 0xf, 0xf, true);
    }
    return r.d;
}
template <int SHIFT>
__device__ __forceinline__ double row_shr_zero(double v)
{
    union { double d; int i[2]; } a, r;
    a.d = v;
    r.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], 0x110 + SHIFT, 0xf, 0xf, true);
    r.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], 0x110 + SHIFT, 0xf, 0xf, true);
    return r.d;
}
__device__ __forceinline__ double read_lane(double v, int lane)
{
    union { double d; int i[2]; } a, r;
    a.d = v;
    r.i[0] = __builtin_amdgcn_readlane(a.i[0], lane);
    r.i[1] = __builtin_amdgcn_readlane(a.i[1], lane);
    return r.d;
}
__device__ __forceinline__ double wave_total(double v)
{
    v += row_shr_zero<1>(v);
    v += row_shr_zero<2>(v);
    v += row_shr_zero<4>(v);
    v += row_shr_zero<8>(v);  // lanes 15, 31, 47, 63 hold their row's total
    return ((read_lane(v, 15) + read_lane(v, 31)) + read_lane(v, 47)) + read_lane(v, 63);
}

__device__ __forceinline__ float lane_shift(float v, int dpp_ctrl_is_shr)
{
    if (dpp_ctrl_is_shr) return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));
}

template <typename T>
struct Neighbours {
    T w, e, n, s;
};
template <typename T>
__device__ __forceinline__ Neighbours<T> fetch(T u, int l_n, int l_s)
{
    Neighbours<T> nb;
    nb.w = lane_shift(u, 1);   // lane - 1
    nb.e = lane_shift(u, 0);   // lane + 1
    nb.n = __shfl(u, l_n, 64); // lane + N
    nb.s = __shfl(u, l_s, 64); // lane - N
    return nb;
}

// all 64 lanes of the wave must call this; lanes >= N*N idle along.  f = F at this lane's point
// (0 for idle lanes).  Returns this lane's U; *iterations_out = number of iterations run.
// T = field type (double, or float in the mixed-precision mode); the norm and the convergence
// test are fp64 either way.
template <typename T>
__device__ __forceinline__ T solve(int N, T h2, T inv, T f, double tol, int max_iter, int *iterations_out)
{
    const int lane = threadIdx.x & 63;
    const int n = N * N;
    const int r = lane / N, c = lane - r * N;
    const bool inside = lane < n && !(r == 0 || c == 0 || r == N - 1 || c == N - 1);
    const int colour = (r + c) & 1;
    const T h2f = h2 * f;
    const int l_s = lane >= N ? lane - N : 0, l_n = lane + N < 64 ? lane + N : 63;
    const double denom = (double)((N - 2) * (N - 2));
    T u = 0.0;  // memset(U, 0)  :993
    Neighbours<T> nb = fetch(u, l_n, l_s);
    int iterations = 0;
    for (;;) {
        // red: U = 0.25*(U[l] + U[r] + U[t] + U[b] - h^2 F)  :1020
        {
            const T nu = T(0.25) * (nb.w + nb.e + nb.n + nb.s - h2f);
            if (inside && colour == 0) u = nu;
        }
        nb = fetch(u, l_n, l_s);
        {   // black :1043
            const T nu = T(0.25) * (nb.w + nb.e + nb.n + nb.s - h2f);
            if (inside && colour == 1) u = nu;
        }
        ++iterations;
        nb = fetch(u, l_n, l_s);  // serves the norm now and the next red pass
        const T rs = inv * (nb.n + nb.s + nb.e + nb.w - 4 * u) - f;  // :560
        const double res = inside ? fabs((double)rs) : 0.0;
        const double err = wave_total(res) / denom;                                             // :1059
        if (!(err > tol) || iterations >= max_iter) break;
    }
    *iterations_out = iterations;
    return u;
}

}  // namespace gsw
}  // namespace k
}  // namespace mg
