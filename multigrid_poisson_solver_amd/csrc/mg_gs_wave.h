// mg_gs_wave.h -- red-black Gauss-Seidel (src/MG_solver_CPU.cpp:952-1066) for a grid of at most
// 64 points by ONE wave: one point per lane, U in a register.  Shared by the stand-alone solver
// kernel (mg_kernels.hip:k_gs_wave) and the coarse-tail kernel (mg_tail.hip).
//
// East/west neighbours are the adjacent lanes (a full-wave DPP shift), north/south are N lanes
// away (ds_bpermute).  The residual norm needs the neighbours of the state after the black
// pass -- exactly what the red pass of the NEXT iteration needs, so each iteration fetches
// neighbours twice, not three times.  The norm is a DPP row scan + 4 readlanes; convergence is
// tested every iteration like the reference's `while (err > target_error)` (:996), with the next
// sweep issued speculatively beside the norm (see solve()).
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
// CTRL: 0x142 = row_bcast:15 (lane 15 of each row -> the next row), 0x143 = row_bcast:31; lanes
// outside ROW_MASK keep `v`
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double row_bcast_add(double v)
{
    union { double d; int i[2]; } a, r;
    a.d = v;
    r.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], CTRL, ROW_MASK, 0xf, false);
    r.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], CTRL, ROW_MASK, 0xf, false);
    return r.d;
}
__device__ __forceinline__ double wave_total(double v)
{
    v += row_shr_zero<1>(v);
    v += row_shr_zero<2>(v);
    v += row_shr_zero<4>(v);
    v += row_shr_zero<8>(v);  // lanes 15, 31, 47, 63 hold their row's total
    // rows 1 and 3 add the total of the row below them, then rows 2-3 add lane 31's (= rows 0+1):
    // lane 63 = (r0 + r1) + (r2 + r3).  Lanes outside the row mask add 0.
    v += row_bcast_add<0x142, 0xa>(v);
    v += row_bcast_add<0x143, 0xc>(v);
    return read_lane(v, 63);
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
    // `sum/denom > tol` decided without the division wherever rounding cannot matter (the fp64
    // division is a dozen instructions of a loop that is bound by its instruction count)
    const double thr = tol * denom, thr_hi = thr * (1.0 + 0x1p-48), thr_lo = thr * (1.0 - 0x1p-48);
    auto above_tol = [&](double sum) {
        if (sum > thr_hi) return true;
        if (sum < thr_lo) return false;
        return sum / denom > tol;  // :1059, :996
    };
    // One sweep = red pass, fetch, black pass, fetch (the second fetch serves the norm of this
    // iterate AND the red pass of the next one).
    auto sweep = [&](T &u, Neighbours<T> &nb) {
        {   // red: U = 0.25*(U[l] + U[r] + U[t] + U[b] - h^2 F)  :1020
            const T nu = T(0.25) * (nb.w + nb.e + nb.n + nb.s - h2f);
            if (inside && colour == 0) u = nu;
        }
        nb = fetch(u, l_n, l_s);
        {   // black :1043
            const T nu = T(0.25) * (nb.w + nb.e + nb.n + nb.s - h2f);
            if (inside && colour == 1) u = nu;
        }
        nb = fetch(u, l_n, l_s);
    };
    T u = 0.0;  // memset(U, 0)  :993
    Neighbours<T> nb = fetch(u, l_n, l_s);
    sweep(u, nb);
    int iterations = 1;
    // The reference tests `err > target` after every sweep (:996).  The norm of iterate i (a
    // residual + a 64-lane reduction, ~40 % of a sweep's dependent chain) does not feed sweep
    // i+1, so both are issued together and sweep i+1 is simply dropped when iterate i turns out
    // to have converged: same iterates, same stopping iteration, shorter critical path.
    for (;;) {
        T un = u;
        Neighbours<T> nbn = nb;
        sweep(un, nbn);                                                                        // speculative
        const T rs = inv * (nb.n + nb.s + nb.e + nb.w - 4 * u) - f;  // :560
        const double res = inside ? fabs((double)rs) : 0.0;
        if (!above_tol(wave_total(res)) || iterations >= max_iter) break;
        u = un;
        nb = nbn;
        ++iterations;
    }
    *iterations_out = iterations;
    return u;
}

}  // namespace gsw
}  // namespace k
}  // namespace mg
