// grid_barrier.hip -- what a device-wide barrier costs inside ONE persistent kernel on MI355X, against the boundary
// between two dependent launches: the price list behind "one persistent kernel for the levels 128...1024"
// (VERDICT r1 item 4).  G co-resident workgroups (at most one per CU) run K rounds of
//     every thread touches its slice (a store and a load of global memory, so that the barrier has to order real data),
//     one thread per workgroup arrives at a monotone counter (device-scope atomic add, release),
//     and spins (device-scope atomic load, acquire; BOUNDED: a lost workgroup ends the kernel with a flag, not a hang).
// Printed: microseconds per round for several G, and the same K rounds as K dependent launches of a kernel with the
// same body.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/grid_barrier.hip -o scripts/ubench/grid_barrier.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void body(double *data, int round)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)gridDim.x * blockDim.x;
    // read what the "next" workgroup wrote in the round before, write mine
    const double v = data[(i + blockDim.x) % n];
    data[i] = v + round;
}

__global__ __launch_bounds__(256) void k_persistent(double *data, unsigned *counter, int *lost, int rounds)
{
    const unsigned G = gridDim.x;
    __shared__ int bail;
    if (threadIdx.x == 0) bail = 0;
    __syncthreads();
    for (int r = 0; r < rounds; ++r) {
        body(data, r);
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(r + 1) * G;
            int polls = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (++polls > 4000000) {   // ~ a second: give up rather than hang the device
                    *lost = 1;
                    bail = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (bail) return;
    }
}

__global__ __launch_bounds__(256) void k_one_round(double *data, int r) { body(data, r); }

int main()
{
    const int rounds = 200;
    double *data;
    unsigned *counter;
    int *lost;
    (void)hipMalloc(&data, (size_t)1024 * 256 * sizeof(double));
    (void)hipMalloc(&counter, sizeof(unsigned));
    (void)hipMalloc(&lost, sizeof(int));
    (void)hipMemset(data, 0, (size_t)1024 * 256 * sizeof(double));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int G : {8, 32, 64, 128, 256}) {
        float best_p = 1e9f, best_l = 1e9f;
        int h_lost = 0;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipMemset(counter, 0, sizeof(unsigned));
            (void)hipMemset(lost, 0, sizeof(int));
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k_persistent, dim3(G), dim3(256), 0, 0, data, counter, lost, rounds);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best_p) best_p = ms;
            (void)hipMemcpy(&h_lost, lost, sizeof(int), hipMemcpyDeviceToHost);
            if (h_lost) break;
            (void)hipEventRecord(e0);
            for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(k_one_round, dim3(G), dim3(256), 0, 0, data, r);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best_l) best_l = ms;
        }
        if (h_lost) {
            printf("G = %3d workgroups: a workgroup never arrived (not co-resident?) -- gave up\n", G);
            continue;
        }
        printf("G = %3d workgroups: %6.2f us per round with a device-wide barrier in one kernel, %6.2f us per round as dependent launches\n", G,
               best_p * 1e3 / rounds, best_l * 1e3 / rounds);
    }
    return 0;
}
