// mg_divconst.h -- correctly rounded division by a constant without the hardware division
// sequence (Markstein): with rc = RN(1/c), two residual corrections of q = x*rc give RN(x/c) --
// the same bits as the IEEE division the reference performs in doProlongation
// (src/MG_solver_CPU.cpp:700 ".../c_dx/c_dx"), in 5 instructions instead of ~14 (fp64).  The FMAs
// are explicit: the translation units are compiled with -ffp-contract=off.  Inputs here are
// ordinary finite numbers.
#pragma once
#include <hip/hip_runtime.h>

namespace mg {
namespace k {

__device__ __forceinline__ double div_by_const(double x, double c, double rc)
{
    double q = x * rc;
    double r = __builtin_fma(-q, c, x);
    q = __builtin_fma(r, rc, q);
    r = __builtin_fma(-q, c, x);
    return __builtin_fma(r, rc, q);
}
__device__ __forceinline__ float div_by_const(float x, float c, float rc)
{
    float q = x * rc;
    float r = __builtin_fmaf(-q, c, x);
    q = __builtin_fmaf(r, rc, q);
    r = __builtin_fmaf(-q, c, x);
    return __builtin_fmaf(r, rc, q);
}

}  // namespace k
}  // namespace mg
