// mg_lane_ops.h -- the small device helpers the node kernels share (mg_stream_impl.h: the wave-streaming smoother of
// the large levels; mg_tile_impl.h: the register-tile smoother of the small ones): neighbour lanes through DPP, the
// exactly-rounded fused forms of the reference's expressions, wave-uniform table reads, bit masks.  Both kernels
// evaluate every point with THESE functions, so they produce the same bits.  Overloads for double and float.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace mg {
namespace k {

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

// `zero ? 0 : v` for a lane constant v whose low 32 bits are zero (0.25, +-1.0, 0.0) and a wave-uniform
// condition: ONE select on the high dword instead of two
__device__ __forceinline__ double uniform_or_zero(bool zero, double v)
{
    const int hi = zero ? 0 : __double2hiint(v);
    return __hiloint2double(hi, 0);
}
__device__ __forceinline__ float uniform_or_zero(bool zero, float v) { return zero ? 0.0f : v; }
__device__ __forceinline__ double fused_mul_add(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fused_mul_add(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// value of lane `lane` (wave-uniform) of a register, as a wave-uniform value
__device__ __forceinline__ int lane_value(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float lane_value(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ double lane_value(double v, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// v with every bit cleared where mask == 0 (mask is 0 or -1): +0.0 or v, also for a NaN
__device__ __forceinline__ double bits_and(double v, int mask) { return __hiloint2double(__double2hiint(v) & mask, __double2loint(v) & mask); }
__device__ __forceinline__ float bits_and(float v, int mask) { return __int_as_float(__float_as_int(v) & mask); }
// the same for a constant whose low 32 bits are zero (0.25, +-1.0): one AND on the high dword
__device__ __forceinline__ double hi_bits_and(double v, int mask) { return __hiloint2double(__double2hiint(v) & mask, 0); }
__device__ __forceinline__ float hi_bits_and(float v, int mask) { return __int_as_float(__float_as_int(v) & mask); }

// sum of a value over the 64 lanes of the wave, through DPP only (no LDS crossbar: the shuffle-based tree costs a lone
// wave ~100 cycles per step): prefix sums inside each row of 16 lanes, then rows 1 and 3 add the total of the row below
// them (row_bcast:15), then rows 2-3 add lane 31's (row_bcast:31); lane 63 holds (r0 + r1) + (r2 + r3)
__device__ __forceinline__ double wave_sum_dpp(double v)
{
    auto shr = [](double x, auto ctrl_tag) {
        constexpr int CTRL = decltype(ctrl_tag)::value;
        return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, true),
                                __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, true));
    };
    v += shr(v, std::integral_constant<int, 0x111>{});
    v += shr(v, std::integral_constant<int, 0x112>{});
    v += shr(v, std::integral_constant<int, 0x114>{});
    v += shr(v, std::integral_constant<int, 0x118>{});
    v += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x142, 0xa, 0xf, false),
                          __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x142, 0xa, 0xf, false));
    v += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x143, 0xc, 0xf, false),
                          __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x143, 0xc, 0xf, false));
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

}  // namespace k
}  // namespace mg
