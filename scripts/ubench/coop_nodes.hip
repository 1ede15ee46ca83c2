// coop_nodes.hip -- the A/B behind "one persistent cooperative kernel for the levels 128/256(/512)" (VERDICT r02, item 1):
// a chain of DEPENDENT small-level nodes, each shaped like a launch of the register-tile kernel (mg_tile_impl.h) --
// a workgroup of 4 waves loads a 24 x 64 window of the previous node's output (so every workgroup reads what OTHER
// workgroups wrote), runs four barrier-separated sweeps on it in registers (boundary rows through LDS), writes its
// 16 x 56 tile -- executed
//   (a) as one launch per node on a stream (what the engine does), and
//   (b) as ONE persistent launch with a device-wide barrier between nodes (monotone counter, device-scope
//       release/acquire, BOUNDED spin: a lost workgroup ends the kernel with a flag, not a hang).
// G = 24 / 80 / 320 workgroups are the tile counts of the levels 128 / 256 / 512.  Printed: microseconds per node.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/coop_nodes.hip -o scripts/ubench/coop_nodes.bin
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int RPW = 6, WAVES = 4, RH = RPW * WAVES, HALO = 4, TY = RH - 2 * HALO, TX = 64 - 2 * HALO;

__device__ __forceinline__ void node(const double *__restrict__ src, double *__restrict__ dst, int N, int tiles_x, int tile, double (*xch)[WAVES][2][64])
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int oy0 = ty * TY, ox0 = tx * TX;
    const int x = ox0 - HALO + lane, yb = oy0 - HALO + wave * RPW;
    const int xc = x < 0 ? 0 : (x < N ? x : N - 1);
    double v[RPW];
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int y = yb + j, yc = y < 0 ? 0 : (y < N ? y : N - 1);
        v[j] = src[(size_t)yc * N + xc];
    }
    int xb = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        xch[xb][wave][0][lane] = v[0];
        xch[xb][wave][1][lane] = v[RPW - 1];
        __syncthreads();
        const double below = wave > 0 ? xch[xb][wave - 1][1][lane] : 0.0;
        const double above = wave < WAVES - 1 ? xch[xb][wave + 1][0][lane] : 0.0;
        xb ^= 1;
        double o[RPW];
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const double so = j > 0 ? v[j > 0 ? j - 1 : 0] : below, nw = j < RPW - 1 ? v[j < RPW - 1 ? j + 1 : 0] : above;
            const double w = __shfl_up(v[j], 1, 64), e = __shfl_down(v[j], 1, 64);
            const double t4 = __builtin_fma(-4.0, v[j], nw + so + e + w) - 1e-3 * v[j];
            o[j] = __builtin_fma(0.25, t4, v[j]);
        }
#pragma unroll
        for (int j = 0; j < RPW; ++j) v[j] = o[j];
    }
    if (x >= ox0 && x < ox0 + TX && x < N) {
#pragma unroll
        for (int j = 0; j < RPW; ++j) {
            const int y = yb + j;
            if (y >= oy0 && y < oy0 + TY && y < N) dst[(size_t)y * N + x] = v[j];
        }
    }
}

__global__ __launch_bounds__(64 * WAVES) void k_one_node(const double *src, double *dst, int N, int tiles_x)
{
    __shared__ double xch[2][WAVES][2][64];
    node(src, dst, N, tiles_x, blockIdx.x, xch);
}

__global__ __launch_bounds__(64 * WAVES) void k_persistent(double *a, double *b, int N, int tiles_x, int nodes, unsigned *counter, int *lost)
{
    __shared__ double xch[2][WAVES][2][64];
    __shared__ int bail;
    const unsigned G = gridDim.x;
    if (threadIdx.x == 0) bail = 0;
    __syncthreads();
    for (int k = 0; k < nodes; ++k) {
        node((k & 1) ? b : a, (k & 1) ? a : b, N, tiles_x, blockIdx.x, xch);
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(k + 1) * G;
            int polls = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (++polls > 4000000) {
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

int main()
{
    const int nodes = 200;
    double *a, *b;
    unsigned *counter;
    int *lost;
    (void)hipMalloc(&a, (size_t)1024 * 1024 * sizeof(double));
    (void)hipMalloc(&b, (size_t)1024 * 1024 * sizeof(double));
    (void)hipMalloc(&counter, sizeof(unsigned));
    (void)hipMalloc(&lost, sizeof(int));
    (void)hipMemset(a, 0, (size_t)1024 * 1024 * sizeof(double));
    (void)hipMemset(b, 0, (size_t)1024 * 1024 * sizeof(double));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipStream_t s;
    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int N : {128, 256, 512}) {
        const int tiles_x = (N + TX - 1) / TX, tiles_y = (N + TY - 1) / TY, G = tiles_x * tiles_y;
        float best_p = 1e9f, best_l = 1e9f;
        int h_lost = 0;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipMemsetAsync(counter, 0, sizeof(unsigned), s);
            (void)hipMemsetAsync(lost, 0, sizeof(int), s);
            (void)hipEventRecord(e0, s);
            hipLaunchKernelGGL(k_persistent, dim3(G), dim3(64 * WAVES), 0, s, a, b, N, tiles_x, nodes, counter, lost);
            (void)hipEventRecord(e1, s);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best_p) best_p = ms;
            (void)hipMemcpy(&h_lost, lost, sizeof(int), hipMemcpyDeviceToHost);
            if (h_lost) break;
            (void)hipEventRecord(e0, s);
            for (int k = 0; k < nodes; ++k)
                hipLaunchKernelGGL(k_one_node, dim3(G), dim3(64 * WAVES), 0, s, (k & 1) ? b : a, (k & 1) ? a : b, N, tiles_x);
            (void)hipEventRecord(e1, s);
            (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep > 0 && ms < best_l) best_l = ms;
        }
        if (h_lost) {
            printf("N = %4d (%3d workgroups): a workgroup never arrived (not co-resident?) -- gave up\n", N, G);
            continue;
        }
        printf("N = %4d (%3d workgroups of 4 waves): %6.2f us per node in ONE persistent kernel with device-wide barriers, %6.2f us per node as dependent launches\n",
               N, G, best_p * 1e3 / nodes, best_l * 1e3 / nodes);
    }
    return 0;
}
