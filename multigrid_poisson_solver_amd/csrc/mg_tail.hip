// mg_tail.hip -- fp64 instantiation of the coarse-tail kernel (source: mg_tail_impl.h)
#define MG_REAL double
#define MG_REAL_NS f64
#include "mg_tail_impl.h"

namespace mg {
namespace k {

// the block exact solver of the tail (mg_tail_impl.h: gauss_seidel_blocks) as a launch of its own, for doExactSolver
// on an even grid of at most 8 x 8 points: F staged into LDS, U written back
__global__ __launch_bounds__(64) void k_gs_blocks(int N, double h2, double inv, double *U, const double *F, double tol, int *state)
{
    const int n = N * N, t = threadIdx.x;
    if (t < n) f64::lds[n + t] = F[t];
    __syncthreads();
    f64::gauss_seidel_blocks(N, h2, inv, 0, n, tol, state);
    if (t < n) U[t] = f64::lds[t];
}
void gauss_seidel_blocks_launch(hipStream_t s, int N, double h2, double inv, double *U, const double *F, double tol, int *state)
{
    hipLaunchKernelGGL(k_gs_blocks, dim3(1), dim3(64), (size_t)2 * N * N * sizeof(double), s, N, h2, inv, U, F, tol, state);
}

bool tail_fits(const TailArgs &a) { return f64::tail_lds_bytes(a) <= (size_t)(160 * 1024 - 512); }
void tail_launch(hipStream_t s, const TailArgs &a, int n_batch, const TailBatchItem *batch_dev) { f64::tail_launch(s, a, n_batch, batch_dev); }

}  // namespace k
}  // namespace mg
