// strip_walk.hip -- what the ACCESS PATTERN of the streaming smoother can reach, without its arithmetic:
// a wave owns a strip of 128 fp64 columns (1 KiB per row, 16 B per lane) and marches down the rows of its chunk,
// reading two arrays and writing one (c = a + b), PF rows in flight.  Against it: the same bytes moved by a
// massive grid of one-shot blocks (scripts/ubench/stream_ceiling.hip reaches 6.6 TB/s that way with nt hints).
// Knobs: rows per chunk (how many concurrent row streams hit each HBM channel), waves per workgroup, lockstep
// (a barrier per row keeps the waves of a workgroup on the same row), non-temporal loads / stores, and the tile
// order (XCD-contiguous as in mg_stream_impl.h, or plain).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/strip_walk.hip -o scripts/ubench/strip_walk.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double2_t __attribute__((ext_vector_type(2)));

template <int PF, bool NT_LD, bool NT_ST, bool LOCKSTEP, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_walk(const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ c,
                                                      int N, int rows_per_chunk, int groups, int n_tiles, int xcd_remap)
{
    int tile = blockIdx.x;
    if (xcd_remap) {
        const int per_xcd = (n_tiles + 7) >> 3;
        tile = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    }
    if (tile >= n_tiles) return;
    const int chunk = tile / groups, group = tile - chunk * groups;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = (group * WAVES + wave) * 128 + lane * 2;
    const bool live = x < N;
    const int y0 = chunk * rows_per_chunk;
    int y1 = y0 + rows_per_chunk;
    if (y1 > N) y1 = N;
    const size_t col = live ? x : 0;
    double2_t ra[PF], rb[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) {
        const int y = y0 + k < y1 ? y0 + k : y1 - 1;
        const size_t off = (size_t)y * N + col;
        ra[k] = NT_LD ? __builtin_nontemporal_load((const double2_t *)(a + off)) : *(const double2_t *)(a + off);
        rb[k] = NT_LD ? __builtin_nontemporal_load((const double2_t *)(b + off)) : *(const double2_t *)(b + off);
    }
    for (int y = y0; y < y1; y += PF) {
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const double2_t va = ra[k], vb = rb[k];
            const int yn = y + k + PF < y1 ? y + k + PF : y1 - 1;
            const size_t offn = (size_t)yn * N + col;
            ra[k] = NT_LD ? __builtin_nontemporal_load((const double2_t *)(a + offn)) : *(const double2_t *)(a + offn);
            rb[k] = NT_LD ? __builtin_nontemporal_load((const double2_t *)(b + offn)) : *(const double2_t *)(b + offn);
            if (y + k < y1 && live) {
                const size_t off = (size_t)(y + k) * N + col;
                if (NT_ST) __builtin_nontemporal_store(va + vb, (double2_t *)(c + off));
                else *(double2_t *)(c + off) = va + vb;
            }
            if (LOCKSTEP) __syncthreads();
        }
    }
}

template <int PF, bool NT_LD, bool NT_ST, bool LOCKSTEP, int WAVES>
static void run(const char *name, const double *a, const double *b, double *c, int N, int rows, int remap)
{
    const int strips = (N + 127) / 128, groups = (strips + WAVES - 1) / WAVES;
    const int chunks = (N + rows - 1) / rows, n_tiles = chunks * groups, grid = (n_tiles + 7) / 8 * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_walk<PF, NT_LD, NT_ST, LOCKSTEP, WAVES>), dim3(grid), dim3(64 * WAVES), 0, 0, a, b, c, N, rows, groups, n_tiles, remap);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    printf("%-44s rows/chunk %5d  tiles %6d  %7.1f us  %6.0f GB/s\n", name, rows, n_tiles, best * 1e3, 24.0 * N * N / (best * 1e-3) / 1e9);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

int main()
{
    const int N = 8192;
    const size_t n = (size_t)N * N;
    double *a, *b, *c;
    hipMalloc(&a, n * 8);
    hipMalloc(&b, n * 8);
    hipMalloc(&c, n * 8);
    hipMemset(a, 0, n * 8);
    hipMemset(b, 0, n * 8);
    hipMemset(c, 0, n * 8);
    for (int rows : {8192, 1024, 293, 128, 32, 8, 2}) {
        run<2, false, true, false, 4>("PF2 ld st.nt 4 waves (the smoother's shape)", a, b, c, N, rows, 1);
        run<2, true, true, false, 4>("PF2 ld.nt st.nt 4 waves", a, b, c, N, rows, 1);
        run<4, true, true, false, 4>("PF4 ld.nt st.nt 4 waves", a, b, c, N, rows, 1);
        run<2, true, true, true, 4>("PF2 ld.nt st.nt 4 waves lockstep", a, b, c, N, rows, 1);
        run<2, true, true, false, 4>("PF2 ld.nt st.nt 4 waves, plain tile order", a, b, c, N, rows, 0);
        run<2, true, true, false, 8>("PF2 ld.nt st.nt 8 waves", a, b, c, N, rows, 1);
        run<2, true, true, false, 16>("PF2 ld.nt st.nt 16 waves", a, b, c, N, rows, 1);
        run<2, true, true, false, 1>("PF2 ld.nt st.nt 1 wave", a, b, c, N, rows, 1);
    }
    return 0;
}
