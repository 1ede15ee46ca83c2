// HBM streaming ceilings for the access mixes of the multigrid kernels (16 B per lane, fp64):
//   copy  (1 read, 1 write), add (2 reads, 1 write), rd2 (2 reads, 0 writes), wr (0 reads, 1 write)
// Build: hipcc --offload-arch=gfx950 -O3 scripts/ubench/stream_ceiling.hip -o scripts/ubench/stream_ceiling.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double2_t __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(const double2_t* __restrict__ a, const double2_t* __restrict__ b, double2_t* __restrict__ c, size_t n, double2_t* sink)
{
    double2_t acc = {0.0, 0.0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (MODE == 0) c[i] = a[i];
        if (MODE == 1) c[i] = a[i] + b[i];
        if (MODE == 2) acc += a[i] + b[i];
        if (MODE == 3) c[i] = acc;
        if (MODE == 4) __builtin_nontemporal_store(a[i] + b[i], &c[i]);
        if (MODE == 5) __builtin_nontemporal_store(__builtin_nontemporal_load(&a[i]), &c[i]);
        if (MODE == 6) __builtin_nontemporal_store(__builtin_nontemporal_load(&a[i]) + __builtin_nontemporal_load(&b[i]), &c[i]);
    }
    if (MODE == 2 && acc.x == 123.456) *sink = acc;
}
int main()
{
    const size_t n = (size_t)8192 * 8192 / 2;  // double2 elements: 512 MiB per array
    double2_t *a, *b, *c, *sink;
    hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&c, n * 16); hipMalloc(&sink, 16);
    hipMemset(a, 0, n * 16); hipMemset(b, 0, n * 16); hipMemset(c, 0, n * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"copy 1R1W", "add 2R1W", "rd2 2R0W", "wr 0R1W", "add 2R1W nt-store", "copy nt-load nt-store", "add nt-loads nt-store"};
    const double bytes[] = {32, 48, 32, 16, 48, 32, 48};
    // grid n/256: one 16 B element per thread, no loop (the shape of a one-row-per-block kernel)
    for (int grid : {2048, 8192, 65536, (int)(n / 256)}) {
        for (int mode = 0; mode < 7; ++mode) {
            float best = 1e9;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) k<0><<<grid, 256>>>(a, b, c, n, sink);
                if (mode == 1) k<1><<<grid, 256>>>(a, b, c, n, sink);
                if (mode == 2) k<2><<<grid, 256>>>(a, b, c, n, sink);
                if (mode == 3) k<3><<<grid, 256>>>(a, b, c, n, sink);
                if (mode == 4) k<4><<<grid, 256>>>(a, b, c, n, sink);
                if (mode == 5) k<5><<<grid, 256>>>(a, b, c, n, sink);
                if (mode == 6) k<6><<<grid, 256>>>(a, b, c, n, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            printf("grid %6d  %-18s %8.1f us  %7.0f GB/s\n", grid, names[mode], best * 1e3, bytes[mode] * (double)n / (best * 1e-3) / 1e9);
        }
    }
    return 0;
}
