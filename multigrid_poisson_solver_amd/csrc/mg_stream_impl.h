// mg_stream_impl.h -- temporally blocked, wave-streaming Jacobi smoother for CDNA4.
//
// doSmoothing (src/MG_solver_CPU.cpp:573-625) runs `step` Jacobi sweeps and then a
// residual-type norm; the driver follows it with getResidual (:268).  Launched sweep by
// sweep that is 24 B of HBM traffic per point and sweep.  This kernel advances S sweeps
// (and, optionally, the residual / error stage) in ONE pass over HBM:
//
//   * a wave64 owns a vertical strip 64*COLS columns wide and marches down the rows of
//     its chunk; lane i holds COLS adjacent columns, so each row is ONE coalesced
//     16 B/lane (COLS = 2) load of U and one of F -- 1 KiB per wave instruction;
//   * time level l of row y needs level l-1 of rows y-1, y, y+1: the wave keeps a
//     two-row history per level in registers and, when input row y arrives, computes
//     level 1 of row y-1, level 2 of row y-2, ... level S of row y-S and finally the
//     residual of row y-S-1 (a software pipeline skewed by one row per level);
//   * the east/west neighbours that live in the adjacent lane come through DPP
//     wave shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1) -- no LDS, no barriers, waves
//     never synchronise;
//   * strips overlap by H = S+1 columns (rounded up to even) and chunks by S+1 rows on
//     each side: the halo is recomputed redundantly instead of exchanged (the overlap
//     re-reads hit L2: the block index is remapped so neighbouring tiles share an XCD);
//   * the next PF rows are always in flight (register FIFO of 2*PF slots, loop unrolled by
//     2*PF; every load is unconditional -- out-of-window rows/columns are clamped -- so the
//     compiler can count the loads in flight and waits with vmcnt(N), not vmcnt(0)).
//
// Two more stages can be fused into the same pass (COLS = 2 builds):
//   * IN_PROLONG: level 0 of a row is U_in + doProlongation(coarse) (:354 + :368),
//     evaluated while the row streams in (coarse rows are cached in registers and
//     rotate as the owner row advances; 1-D weights come from the host tables);
//   * RESTRICT: the signed residual rows are restricted on the fly into the next
//     level's F (:287) -- the lane whose column pair holds lo[cc] combines its pair (or
//     its right neighbour's, via one DPP shift) of two consecutive residual rows; the
//     residual D itself then never touches HBM.
// With both, a V-cycle level costs  F + U + F_coarse (18 B/point) on the way down and
// U + coarse + F + U (26 B/point) on the way up.
//   * PRE > 0 (the fused `1` node of the big levels): the U the `-1` node smoothed from zero is not read -- that node
//     did not even store it (no_out) -- but recomputed in the same pass: PRE sweeps from the zero field on the same F
//     (levels 1..PRE), the prolongation added to level PRE, S more sweeps.  The level pair then costs
//     F + F_coarse (10 B/point) down and coarse + F + U (18 B/point) up, for PRE more sweeps of arithmetic.
//
// Every point is updated with exactly the reference's expression and association order
// (-ffp-contract=off), so the result is bit-identical to S separate sweeps; a halo point
// computed twice gets the same bits twice.  Algorithmic traffic of S sweeps + residual:
// read U, read F, write U, write D = 32 B per point instead of 24*S + 24 + 8.
//
// This file is the kernel source for BOTH field types: it is included once with MG_REAL = double
// (mg_stream.hip, the fp64 path whose results are bit-identical to the reference) and once with
// MG_REAL = float (mg_stream_f32.hip, the fp32 smoother of the mixed-precision mode).  With
// MG_REAL = double every expression is exactly what it was before the split.
#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include <algorithm>
#include <cstdlib>
#include <vector>

#include "mg_divconst.h"
#include "mg_internal.h"
#include "mg_lane_ops.h"

#if !defined(MG_REAL) || !defined(MG_REAL_NS)
#error "define MG_REAL (double|float) and MG_REAL_NS (f64|f32) before including mg_stream_impl.h"
#endif

namespace mg {
namespace k {
namespace MG_REAL_NS {

typedef MG_REAL real_t;

// host-built transfer tables in the field type (weights) -- see mg_tables.cpp
struct StreamTables {
    const int *p_orow = nullptr, *p_ocol = nullptr;
    const real_t *p_rhi = nullptr, *p_rlo = nullptr, *p_chi = nullptr, *p_clo = nullptr;
    real_t c_dx = 0, c_dx_rcp = 0;
    bool p_cols4_ok = false;   // ProlongTable::fusable4
    const int *r_inv = nullptr;
    const real_t *r_w = nullptr, *r_wf = nullptr;
};


constexpr int PF_DEFAULT = 2;    // rows of U and F in flight per lane (FIFO of 2*PF slots)
#ifndef MG_WAVES_PER_WG
#define MG_WAVES_PER_WG 4        // (experiment switch: scripts/build_variant.sh -DMG_WAVES_PER_WG=1|2|8)
#endif
constexpr int WAVES_PER_WG = MG_WAVES_PER_WG;  // 4 adjacent strips of one chunk
constexpr int MAX_S = 4;
#ifndef MG_PF
#define MG_PF 2   // prefetch depth of this build (scripts/build_variant.sh: -DMG_PF=3)
#endif

// The recomputing fused `1` node (PRE sweeps from zero redone in flight + S post-smoothing sweeps, template parameter
// PRE of k_jacobi_stream) exists for every pair whose pipeline of L = PRE + S levels is at most 6 deep: V(1,1), V(2,2),
// V(3,3) of the fixed-step files and the unequal pairs a file with per-node step counts (con_step = 0) can ask for
// (1+2, 2+1, 1+3, 3+1, 2+3, 3+2, 1+4, 4+1, 2+4, 4+2).  Six levels is what the F ring (8 rows in LDS, a row lives L + 2
// steps) and two waves per SIMD hold; V(4,4) would be 8 levels and stays store/re-read.  Everybody who decides to drop a
// level's U (recompute_available, the slab schedule) asks this function, so a variant build can never drop a field it
// cannot make again.
constexpr bool recompute_instantiated(int pre, int steps)
{
    return MG_PF == 2 && pre >= 1 && steps >= 1 && pre <= MAX_S && steps <= MAX_S && pre + steps <= 6;
}

enum InMode { IN_LOAD = 0, IN_ZERO = 1, IN_PROLONG = 2 };

typedef real_t real2_t __attribute__((ext_vector_type(2)));
typedef real_t real4_t __attribute__((ext_vector_type(4)));

struct StreamParams {
    int N;
    real_t dx2, inv;
    const real_t *in;   // IN_ZERO: unused
    const real_t *F;
    real_t *out;
    real_t *D;          // nullptr: residual not stored
    int d_sign;
    double *part;       // nullptr: no error norm; else one partial per wave
    long long *trace;   // MG_STREAM_TRACE builds only
    int cols4;          // fp32 fields: 4 columns (16 B) per lane instead of 2
    int rows_per_chunk;
    int groups;         // workgroups per chunk row
    int n_blocks;       // chunks * groups
    // row window (1-D row-slab decomposition): the fine arrays in/F/out/D start at global
    // row row_base and hold rows_local rows; this launch updates rows [own_y0, own_y1).
    // Single GPU: row_base = 0, rows_local = N, own = [0, N).
    int row_base, rows_local, own_y0, own_y1;
    int norm_y0, norm_y1;           // rows counted in the error norm (a subset of the updated rows)
    double *out_wide;               // fp32 fields only: store the result as fp64 here instead of `out`
    int coarse_base, coarse_rows;   // IN_PROLONG: window of the coarse array
    int raw_norm;                   // error output is the raw sum over the owned rows
    int nt_min_n;                   // fused `1` nodes of grids at least this large store U non-temporally
    int no_out;                     // the smoothed U is not stored (a `-1` node whose `1` node recomputes it: PRE)
    int pre;                        // fused `1` node: sweeps from zero to recompute instead of reading `in`
    int fc_base, fc_rows;           // RESTRICT: window of Fc (global row of its first local row, rows)
    // IN_PROLONG: coarse grid and the host-built tables of doProlongation
    const real_t *coarse;
    int Nc;
    const int *p_orow, *p_ocol;
    const real_t *p_rhi, *p_rlo, *p_chi, *p_clo;
    real_t c_dx, c_dx_rcp;   // coarse spacing and RN(1/c_dx)
    // RESTRICT: next level's F and the host-built tables of doRestriction
    real_t *Fc;
    int M;
    const int *r_inv;   // [N] fine index -> coarse index whose lower-left sample it is, or -1
    const real_t *r_w;  // [M] weights by coarse index
    const real_t *r_wf; // [N] r_w[r_inv[x]] by fine index (0 where r_inv < 0)
    // A BATCH of instances of the same node (the independent visits of one level in a W-cycle, mg_cycle.cpp): blockIdx.y
    // picks the instance, whose arrays replace in / F / out / Fc / coarse; its norm partials follow those of the
    // instance before it.  nullptr: the one instance described above.
    const NodeBatchItem *batch;
    int part_stride;    // partials per instance
    int n_batch;                // (host side only: instances of the launch, and where each one's error goes)
    double *const *err_outs;
};


// Read-only host-built tables are read through the constant address space: the compiler may
// then use scalar loads (s_load, SGPR result, lgkmcnt) for wave-uniform indices instead of
// vector loads that would queue behind the streaming loads in vmcnt order.
typedef const int __attribute__((address_space(4))) *const_int_ptr;
typedef const real_t __attribute__((address_space(4))) *const_real_ptr;
__device__ __forceinline__ int table_i(const int *t, int i) { return ((const_int_ptr)(uintptr_t)t)[i]; }
__device__ __forceinline__ real_t table_d(const real_t *t, int i) { return ((const_real_ptr)(uintptr_t)t)[i]; }

template <int COLS>
struct Row {
    real_t v[COLS];
};

// Row loads are UNCONDITIONAL: rows and columns outside the window are clamped to a valid
// address instead of being skipped.  What such a load returns is never consumed by a point
// that is stored (a rim point keeps its value and never looks at its neighbours), and a load
// that is always issued lets the compiler count the loads in flight: with predicated loads it
// has to assume none was issued and waits for vmcnt(0) at every use.
// Addresses are a wave-uniform row pointer (SGPR pair, advanced by one row per step) plus a 32-bit
// per-lane element offset: the global_load/store then takes its `saddr + voffset` form and the row
// walk costs two scalar additions instead of 64-bit vector arithmetic per lane.
// (the lane offset is in BYTES: `uniform pointer + zero-extended 32-bit value` is the shape the instruction selector
// needs; an element index would have to be scaled in 64 bits first)
#ifndef MG_ZERO_LEVEL1
#define MG_ZERO_LEVEL1 1   // 0: the first sweep from zero through the general expression (A/B switch)
#endif
#ifndef MG_NT_LOADS
#define MG_NT_LOADS 0   // experiment: 1 = F rows, 2 = U and F rows loaded non-temporally
#endif
template <int COLS, bool NT = false>
__device__ __forceinline__ Row<COLS> load_row(const real_t *__restrict__ row, unsigned col_bytes)
{
    Row<COLS> r;
    const char *a = reinterpret_cast<const char *>(row) + col_bytes;
    if constexpr (COLS == 2) {
        const real2_t t = NT ? __builtin_nontemporal_load(reinterpret_cast<const real2_t *>(a)) : *reinterpret_cast<const real2_t *>(a);
        r.v[0] = t.x;
        r.v[1] = t.y;
    } else if constexpr (COLS == 4) {
        const real4_t t = NT ? __builtin_nontemporal_load(reinterpret_cast<const real4_t *>(a)) : *reinterpret_cast<const real4_t *>(a);
        r.v[0] = t.x;
        r.v[1] = t.y;
        r.v[2] = t.z;
        r.v[3] = t.w;
    } else {
        r.v[0] = *reinterpret_cast<const real_t *>(a);
    }
    return r;
}

// NT: the array is far larger than the caches and is next read by a later kernel: a non-temporal store keeps it from
// displacing the rows the neighbouring tiles re-read.  A COMPILE-TIME property of the kernel: as a run-time flag
// (`if (nt) nontemporal store; else store`) the two branches are merged into one plain store by the compiler and the
// hint is silently lost -- which is what the round-1/2 kernels ran with until this was noticed in the ISA.
template <int COLS, bool NT>
__device__ __forceinline__ void store_row(real_t *__restrict__ row, unsigned col_bytes, const Row<COLS> &r)
{
    char *a = reinterpret_cast<char *>(row) + col_bytes;
    if constexpr (COLS == 2) {
        real2_t t;
        t.x = r.v[0];
        t.y = r.v[1];
        if constexpr (NT) __builtin_nontemporal_store(t, reinterpret_cast<real2_t *>(a));
        else *reinterpret_cast<real2_t *>(a) = t;
    } else if constexpr (COLS == 4) {
        real4_t t;
        t.x = r.v[0];
        t.y = r.v[1];
        t.z = r.v[2];
        t.w = r.v[3];
        if constexpr (NT) __builtin_nontemporal_store(t, reinterpret_cast<real4_t *>(a));
        else *reinterpret_cast<real4_t *>(a) = t;
    } else {
        *reinterpret_cast<real_t *>(a) = r.v[0];
    }
}

// halo columns per side: level S must be valid one column beyond the owned strip for the
// residual stage, two for the fused restriction (it reads the residual one column and one
// row beyond the owned tile); rounded up to even so strips stay 16 B aligned
template <int S, bool RESTRICT, int COLS = 2>
struct Halo {
    static constexpr int A = COLS > 2 ? COLS : 2;  // strips start on a multiple of the lane's column count (16 B loads)
    static constexpr int value = (S + (RESTRICT ? 2 : 1) + A - 1) / A * A;
};


// NCV consecutive coarse values of one coarse row, starting at this lane's base column (COLS/2 + 2: the COLS fine
// columns of a lane lie in at most COLS/2 + 1 coarse cells)
template <int NCV>
struct CoarseV {
    real_t v[NCV];
};
// unconditional like load_row: rows fetched ahead of need may lie outside the local window (clamped)
// and columns past the grid are clamped (never consumed); col[] are the lane's three clamped columns
template <int NCV>
__device__ __forceinline__ CoarseV<NCV> load_coarse(const real_t *__restrict__ coarse, int Nc, int base, int rows, int row,
                                                    const unsigned (&col)[NCV])
{
    CoarseV<NCV> c;
    int r = row - base;
    r = r < 0 ? 0 : (r < rows - 1 ? r : rows - 1);
    const char *b = reinterpret_cast<const char *>(coarse + (size_t)r * Nc);
#pragma unroll
    for (int q = 0; q < NCV; ++q) c.v[q] = *reinterpret_cast<const real_t *>(b + col[q]);   // col[] in bytes
    return c;
}


// the same with the local (already clamped) row index of the coarse array
template <int NCV>
__device__ __forceinline__ CoarseV<NCV> load_coarse_local(const real_t *__restrict__ coarse, int Nc, int local_row, const unsigned (&col)[NCV])
{
    CoarseV<NCV> c;
    const char *b = reinterpret_cast<const char *>(coarse + (size_t)local_row * Nc);
#pragma unroll
    for (int q = 0; q < NCV; ++q) c.v[q] = *reinterpret_cast<const real_t *>(b + col[q]);
    return c;
}

// f(integral_constant<int, 0>), f(<1>), ... while f returns true
template <class F, int... K>
__device__ __forceinline__ void unrolled_while(F &&f, std::integer_sequence<int, K...>)
{
    (void)(f(std::integral_constant<int, K>{}) && ...);
}

// waves per SIMD the register allocator is asked to make room for (experiment switch MG_WPE_DOWN: the fused `-1` node
// of fp64 fields needs 131-135 VGPRs -- three waves per SIMD, four from 128 down)
#ifndef MG_WPE_DOWN
#define MG_WPE_DOWN 0
#endif
template <int S, int COLS, int IN, bool RESTRICT, int PRE>
struct WavesPerSimd {
    static constexpr int min = (MG_WPE_DOWN > 0 && sizeof(real_t) == 8 && COLS == 2 && IN == 1 /* IN_ZERO */ && RESTRICT && S <= 3) ? MG_WPE_DOWN : 1;
};

// Round 4: the recomputing `1` node keeps its F rows in LDS instead of in a register ring (MG_LDS_RING, A/B switch).
// A row of F lives PF + L + 2 = 10 steps; as registers that was a ring of 16 slots = 64 VGPRs of the 249 that held the
// kernel to two waves per SIMD with the LDS of the CU idle.  Now a row is loaded PF steps ahead into a short register
// FIFO and, when it arrives, goes to the wave's own slice of LDS TWICE: dx2*F (what the L sweeps consume: the product
// is formed once per point instead of once per point and sweep -- the same rounding, hence the same bits) in a ring of 8
// rows x 16 B per lane, and the ONE column of the row the error norm will look at (see HALF below) in a ring of 8 rows
// x 8 B per lane: 12 KiB per wave, 144 KiB for the 12 waves of a CU at three waves per SIMD.  No barrier: a wave only
// ever touches its own slice, and the LDS operations of one wave execute in order.  The ring slots are compile-time
// offsets of the ds_read/ds_write instructions (the loop body covers 8 steps = one turn of the ring).
// HALF: doSmoothing's error norm (src/MG_solver_CPU.cpp:610/:617) counts the interior points with (row + col) even only,
// and this node stores no residual, so the residual stage forms ONE column per row: the pipeline starts on an even row
// (a chunk that begins on an odd row marches one row more), which makes the parity of every row a compile-time property
// of its position in the loop body.  The sum gets the very terms it got before (the masked column contributed +0.0).
#ifndef MG_LDS_RING
#define MG_LDS_RING 1
#endif
// PACKED fp32 arithmetic (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: two fp32 operations per lane and instruction, the
// only way to the fp32 rate of the vector unit): the fp32 kernels evaluate a lane's column pairs (0,1) [and (2,3)] with
// them.  Every element still sees the reference's operations in the reference's order (each
// packed instruction rounds its two elements like the scalar one), so the numpy restatement still pins the bits.
#ifndef MG_PACKED_F32
#define MG_PACKED_F32 1
#endif
typedef float pk_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pk_t pk(float a, float b)
{
    pk_t r;
    r.x = a;
    r.y = b;
    return r;
}
__device__ __forceinline__ pk_t pk_fma(pk_t a, pk_t b, pk_t c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ pk_t pk_div_by_const(pk_t x, float c, float rc)   // div_by_const (mg_divconst.h) on two elements
{
    const pk_t cc = pk(c, c), rr = pk(rc, rc);
    pk_t q = x * rr;
    pk_t r = pk_fma(-q, cc, x);
    q = pk_fma(r, rr, q);
    r = pk_fma(-q, cc, x);
    return pk_fma(r, rr, q);
}
template <int COLS, int PRE>
struct LdsRing {
    // (fp32 fields with 4 columns -- 16 B -- per lane as well: the ring is what makes that form of the node fit two waves per SIMD)
    static constexpr bool value = MG_LDS_RING && PRE > 0 && (COLS == 2 || (COLS == 4 && sizeof(real_t) == 4));
};
// a lane's COLS columns of one row as one LDS access, and the half of them the error norm counts in a row
template <int COLS> struct RingVec { typedef real_t full; typedef real_t half; };
template <> struct RingVec<2> { typedef real2_t full; typedef real_t half; };
template <> struct RingVec<4> { typedef real4_t full; typedef real2_t half; };
template <int COLS>
__device__ __forceinline__ typename RingVec<COLS>::full ring_pack(const Row<COLS> &r)
{
    typename RingVec<COLS>::full t;
    if constexpr (COLS == 1) t = r.v[0];
    else {
#pragma unroll
        for (int j = 0; j < COLS; ++j) t[j] = r.v[j];
    }
    return t;
}
template <int COLS>
__device__ __forceinline__ Row<COLS> ring_unpack(const typename RingVec<COLS>::full &t)
{
    Row<COLS> r;
    if constexpr (COLS == 1) r.v[0] = t;
    else {
#pragma unroll
        for (int j = 0; j < COLS; ++j) r.v[j] = t[j];
    }
    return r;
}
#ifndef MG_PF_UP
#define MG_PF_UP 3    // rows of F (and coarse rows) in flight per lane in the LDS-ring form of the `1` node
#endif
#ifndef MG_PF_DOWN
#define MG_PF_DOWN 3  // (3: 151 -> 147.5 us at 8192, same 131 VGPRs)  rows of F in flight per lane in the zero-start `-1` node
#endif
#ifndef MG_WPE_UP
#define MG_WPE_UP 2   // (3 = 168 VGPRs: the allocator spills 15-27 dwords, and a spill reload drains the load queue: 580 us)   waves per SIMD the allocator is asked to make room for in the LDS-ring form of the `1` node
#endif

template <int S, int COLS, int IN, bool RESTRICT, int PF = PF_DEFAULT, bool NT = false, int PRE = 0>
__global__ __launch_bounds__(64 * WAVES_PER_WG) __attribute__((amdgpu_waves_per_eu(LdsRing<COLS, PRE>::value ? MG_WPE_UP : WavesPerSimd<S, COLS, IN, RESTRICT, PRE>::min)))
void k_jacobi_stream(const StreamParams p)
{
    static_assert(COLS == 2 || COLS == 4 || (IN != IN_PROLONG && !RESTRICT), "fused transfer stages need column pairs");
    static_assert(COLS != 4 || sizeof(real_t) == 4, "4 columns per lane = one 16-byte access: fp32 fields only");
    constexpr int NCV = COLS / 2 + 2;  // coarse values a lane needs of one coarse row
    constexpr int NS = COLS >= 2 ? COLS / 2 : 1;  // coarse columns a lane can produce (one per column pair)
    // PRE > 0 (fused `1` node only): the level's pre-smoothed U is NOT read -- the matching `-1` node did not store it --
    // but recomputed: PRE sweeps from the zero field on the same F (levels 1..PRE, the very expressions of the `-1`
    // node), the prolongation is added to level PRE, S more sweeps follow (levels PRE+1..L).  8 B per point less to
    // read here, 8 B per point less to write there, for PRE more sweeps of arithmetic in a kernel that waits for memory.
    static_assert(PRE == 0 || (IN == IN_PROLONG && !RESTRICT), "recomputed pre-smoothing belongs to the fused `1` node");
    constexpr int L = S + PRE;                         // levels of the pipeline
    constexpr bool LDSR = LdsRing<COLS, PRE>::value;   // F rows in LDS (dx2*F for the sweeps, one column of F for the norm)
    constexpr bool HALF = LDSR;                        // norm-only residual stage: one column per row
    constexpr bool PACKED = MG_PACKED_F32 && (COLS == 2 || COLS == 4) && sizeof(real_t) == 4;   // column pairs through packed fp32 instructions
#ifndef MG_TB_DIRECT
#define MG_TB_DIRECT 0
#endif
    constexpr bool TB_DIRECT = LDSR && MG_TB_DIRECT;   // table blocks loaded where they take over (no second set of registers)
    constexpr int NB = LDSR ? 8 : (PF + L + 2 <= 8 ? 8 : 16);  // slots of the F ring = row steps of the loop body
    constexpr int NFR = LDSR ? (PF < 4 ? 4 : 8) : NB;  // F rows held in registers (LDSR: the rows in flight only)
    static_assert((LDSR ? L + 2 <= NB : PF + L + 2 <= NB) && PF < 8, "the F ring has NB slots, the U ring 4 or 8");
    static_assert(!LDSR || (IN == IN_PROLONG && !RESTRICT), "the LDS ring belongs to the recomputing `1` node");
    typedef typename RingVec<COLS>::full ring_t;
    typedef typename RingVec<COLS>::half ringh_t;
    __shared__ ring_t s_g[LDSR ? WAVES_PER_WG : 1][LDSR ? NB : 1][LDSR ? 64 : 1];    // dx2*F, rows yin-L .. yin
    __shared__ ringh_t s_f[LDSR ? WAVES_PER_WG : 1][LDSR ? NB : 1][LDSR ? 64 : 1];   // F of the columns the norm counts
    constexpr int W = 64 * COLS;
    constexpr int H = Halo<L, RESTRICT, COLS>::value;
    constexpr int OW = W - 2 * H;  // columns a wave owns

    // blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous
    // range of tiles so halo re-reads hit its L2.  Speed only, never correctness.
#ifdef MG_STREAM_TRACE   // diagnostics: where a small launch spends its time (tile 0, wave 0 -> p.D as a scratch of 4 doubles)
    const long long tr0 = wall_clock64();
#endif
    const int per_xcd = (p.n_blocks + 7) >> 3;
    const int tile = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (tile >= p.n_blocks) return;
    // the arrays of this instance (wave-uniform; a batch reads them through the constant address space: scalar loads)
    const real_t *a_in = p.in, *a_F = p.F, *a_coarse = p.coarse;
    real_t *a_out = p.out, *a_Fc = p.Fc;
    double *a_part = p.part;
    if (p.batch) {
        typedef const NodeBatchItem __attribute__((address_space(4))) *item_ptr;
        const item_ptr b = (item_ptr)(uintptr_t)(p.batch + blockIdx.y);
        a_in = static_cast<const real_t *>(b->in);
        a_F = static_cast<const real_t *>(b->F);
        a_coarse = static_cast<const real_t *>(b->coarse);
        a_out = static_cast<real_t *>(b->out);
        a_Fc = static_cast<real_t *>(b->Fc);
        if (a_part) a_part += (size_t)blockIdx.y * (size_t)p.part_stride;
    }
    const int chunk = tile / p.groups;
    const int group = tile - chunk * p.groups;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int strip = group * WAVES_PER_WG + wave;
    const int N = p.N;
    const int own_x0 = strip * OW;
    if (own_x0 >= N) {  // no barriers in this kernel: a wave may leave at any time
        if (a_part && lane == 0) a_part[(size_t)tile * WAVES_PER_WG + wave] = 0.0;
        return;
    }
    const int y0 = p.own_y0 + chunk * p.rows_per_chunk;
    int y1 = y0 + p.rows_per_chunk;
    if (y1 > p.own_y1) y1 = p.own_y1;
    if (y0 >= p.own_y1) return;
    // rows that exist in the local window and in the grid
    const int av_lo = p.row_base > 0 ? p.row_base : 0;
    const int av_hi = p.row_base + p.rows_local < N ? p.row_base + p.rows_local : N;

    const int xl = own_x0 - H + lane * COLS;  // this lane's first column
    bool col_in[COLS], col_edge[COLS];
    bool lane_owns = true;
    // rim columns keep their value: the update is U + q*t with q = 0.25 inside and 0 on the rim
    real_t qc[COLS];
#pragma unroll
    for (int j = 0; j < COLS; ++j) {
        const int x = xl + j;
        col_in[j] = x >= 0 && x < N;
        col_edge[j] = x <= 0 || x >= N - 1;
        lane_owns = lane_owns && x >= own_x0 && x < own_x0 + OW && x < N;
        qc[j] = col_edge[j] ? real_t(0.0) : real_t(0.25);
    }
    // error norm (:610/:617): interior points with (row + col) even, each counted by the lane that owns it: as bit
    // masks per lane and row parity (AND-ed onto |r|, so a NaN in a row that is not counted cannot leak in)
    int nm_even[COLS], nm_odd[COLS];
#pragma unroll
    for (int j = 0; j < COLS; ++j) {
        const bool mine = lane_owns && !col_edge[j];
        nm_even[j] = (mine && ((xl + j) & 1) == 0) ? -1 : 0;  // even rows count even columns
        nm_odd[j] = (mine && ((xl + j) & 1) != 0) ? -1 : 0;
    }
    const bool lane_loads = col_in[0] && col_in[COLS - 1];  // COLS == 2: N even, xl even
    const real_t dx2 = p.dx2, inv = p.inv;
    const bool want_res = RESTRICT || p.D != nullptr || p.part != nullptr;
    // clamped column of this lane's loads (lanes left/right of the grid re-read a valid pair); lane offsets in
    // bytes, see load_row
    const unsigned col_off = (unsigned)(xl < 0 ? 0 : (xl > N - COLS ? N - COLS : xl)) * (unsigned)sizeof(real_t);
    const unsigned col_st = (unsigned)(xl < 0 ? 0 : xl) * (unsigned)sizeof(real_t);  // stores are predicated on lane_owns: xl is in range there
    const unsigned row_bytes = (unsigned)N * (unsigned)sizeof(real_t);

    // ---- fused prolongation input: per-lane column tables, horizontal interpolants of two coarse rows
    unsigned pc_col[NCV];               // the lane's coarse columns (clamped), as byte offsets
    int pc_cell[COLS];                  // coarse cell of fine column j, relative to the lane's first cell
    real_t pc_hi[COLS], pc_lo[COLS];
    // hA[j] / hB[j]: doProlongation's (c1*(c2x-f_x) + c2*(f_x-c1x)) resp. (c3*... + c4*...) of :700 for coarse
    // rows c_row / c_row + 1 at this lane's fine column j.  They change only when the owner row advances
    // (every other fine row), not with every fine row.
    real_t hA[COLS], hB[COLS];
#pragma unroll
    for (int q = 0; q < NCV; ++q) pc_col[q] = 0u;
#pragma unroll
    for (int j = 0; j < COLS; ++j) {
        pc_cell[j] = 0;
        pc_hi[j] = pc_lo[j] = hA[j] = hB[j] = real_t(0.0);
    }
    int c_row = -0x40000000;            // coarse row behind hA (wave-uniform); hB belongs to c_row + 1
    if constexpr (IN == IN_PROLONG) {
        if (lane_loads) {
            const int pc_base = p.p_ocol[xl];
            const int last = p.Nc - 1;
#pragma unroll
            for (int j = 0; j < COLS; ++j) {
                pc_cell[j] = p.p_ocol[xl + j] - pc_base;
                pc_hi[j] = p.p_chi[xl + j];
                pc_lo[j] = p.p_clo[xl + j];
            }
#pragma unroll
            for (int q = 0; q < NCV; ++q) pc_col[q] = (unsigned)(pc_base + q < last ? pc_base + q : last) * (unsigned)sizeof(real_t);
        }
    }
    auto interpolate = [&](const CoarseV<NCV> &c, real_t (&h)[COLS]) {
#pragma unroll
        for (int j = 0; j < COLS; ++j) {
            real_t ca = c.v[0], cb = c.v[1];
#pragma unroll
            for (int q = 1; q + 1 < NCV; ++q) {  // (host-checked: the cell index stays below NCV - 1)
                const bool here = pc_cell[j] == q;
                ca = here ? c.v[q] : ca;
                cb = here ? c.v[q + 1] : cb;
            }
            h[j] = ca * pc_hi[j] + cb * pc_lo[j];
        }
    };

    // ---- fused restriction output: which coarse column this lane produces -------------
    // (one slot per column pair of the lane: lo[] advances by >= 2, so a pair holds at most one sample)
    int rc_col[NS];           // coarse column (interior) or -1
    bool rc_shift[NS];        // its lower-left fine sample is the pair's SECOND column
    real_t rw_a[NS], rw_b[NS];
    Row<COLS> d_prev;         // signed residual of the previous row
#pragma unroll
    for (int j = 0; j < COLS; ++j) d_prev.v[j] = 0.0;
#pragma unroll
    for (int q = 0; q < NS; ++q) {
        rc_col[q] = -1;
        rc_shift[q] = false;
        rw_a[q] = rw_b[q] = real_t(0.0);
    }
    if constexpr (RESTRICT) {
        if (lane_owns) {
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                const int ca = p.r_inv[xl + 2 * q], cb = p.r_inv[xl + 2 * q + 1];
                rc_col[q] = ca >= 0 ? ca : cb;
                rc_shift[q] = ca < 0 && cb >= 0;
                if (rc_col[q] >= 0) {
                    rw_a[q] = p.r_w[rc_col[q]];
                    rw_b[q] = real_t(1.0) - rw_a[q];  // src/MG_solver_CPU.cpp:665
                }
            }
        }
    }
    // the rim of the next level's F is zero (doRestriction's memset, :651): rim columns are
    // written by the lanes that own fine columns 0 and N-1 alongside every coarse row, rim rows
    // by the chunks that hold fine rows 0 and N-1
    const bool first_col_lane = RESTRICT && lane_owns && xl == 0;
    const bool last_col_lane = RESTRICT && lane_owns && xl + COLS == N;
    if constexpr (RESTRICT) {
        for (int edge = 0; edge < 2; ++edge) {
            if (edge == 0 ? (y0 != 0) : (y1 != N)) continue;
            real_t *row = a_Fc + (size_t)((edge == 0 ? 0 : p.M - 1) - p.fc_base) * p.M;
#pragma unroll
            for (int q = 0; q < NS; ++q)
                if (rc_col[q] >= 0) row[rc_col[q]] = 0.0;
            if (first_col_lane) row[0] = 0.0;
            if (last_col_lane) row[p.M - 1] = 0.0;
        }
    }
    // the signed residual that feeds the fused restriction: r * (+-1 inside, 0 on the rim) -- the products are
    // exact, and the sign of a rim zero cannot reach a coarse value (it is added to nonzero terms or to zeros)
    real_t ms[COLS];
#pragma unroll
    for (int j = 0; j < COLS; ++j) ms[j] = col_edge[j] ? real_t(0.0) : (p.d_sign < 0 ? real_t(-1.0) : real_t(1.0));

    // Lanes that hold column pairs start on an even column (xl is even), and the march starts on an even row (a chunk
    // that begins on an odd row takes one step more): the parity of every row is then a compile-time property of its
    // position in the loop body, and so is WHICH of a lane's columns the error norm counts in it ((row + col) even).
    constexpr bool EVEN_START = COLS >= 2;
    const int y0e = EVEN_START ? (y0 & ~1) : y0;
    const int y_first = y0e - (L + 1);                            // first input row
    const int T = (y1 - y0e) + 2 * (L + 1) + (RESTRICT ? 1 : 0);  // input rows consumed
    const int y_end = y_first + T;                                // one past the last input row

    // ---- wave-uniform per-row table entries: ONE REGISTER PER TABLE, one row per lane.  Lane L of a block holds the
    // entry of row (first row of the block) + L, a step picks its row with v_readlane; a block lasts 64 steps
    // and the next one is loaded 8 steps before it is needed.  (Scalar loads per row -- with their guards, their
    // address arithmetic and, above all, a FIFO of SGPR pairs that spilled -- cost far more instructions.)
    //   prolongation: owner coarse row (needed PF steps ahead, for the coarse prefetch) and the two row weights;
    //   restriction: coarse row sampled at fine row (input row - S - 2) and its weight c.
    int tb_own = -1, tbn_own = -1;                          // block of the row being PREFETCHED (step + PF)
    real_t tb_yh = 0.0, tb_yl = 0.0, tbn_yh = 0.0, tbn_yl = 0.0;  // block of the row being consumed
    int tb_rc = -1, tbn_rc = -1;
    real_t tb_rw = 0.0, tbn_rw = 0.0;
    auto load_own_block = [&](int first_row) {  // -> tbn_own
        const int y = first_row + lane;
        const bool ok = y >= av_lo && y < av_hi && y < y_end;
        const int yc = y < 0 ? 0 : (y < N ? y : N - 1);
        const int v = p.p_orow[yc];
        tbn_own = ok ? v : -1;
    };
    auto load_weight_block = [&](int first_row) {  // -> tbn_yh, tbn_yl
        const int y = first_row + lane;
        const int yc = y < 0 ? 0 : (y < N ? y : N - 1);
        tbn_yh = p.p_rhi[yc];
        tbn_yl = p.p_rlo[yc];
    };
    auto load_restrict_block = [&](int first_row) {  // -> tbn_rc, tbn_rw; rows outside the chunk sample nothing
        const int y = first_row + lane;
        const bool ok = y >= y0 && y < y1;
        const int yc = y < 0 ? 0 : (y < N ? y : N - 1);
        const int v = p.r_inv[yc];
        const real_t w = p.r_wf[yc];
        tbn_rc = ok ? v : -1;
        tbn_rw = w;
    };
    // the prolongation lands on row (input row - PRE): its tables (and the coarse rows that travel with the F rows) are
    // those of that row
    const int yp_first = y_first - PRE;
    if constexpr (IN == IN_PROLONG) {
        load_own_block(yp_first);
        tb_own = tbn_own;
        load_weight_block(yp_first);
        tb_yh = tbn_yh;
        tb_yl = tbn_yl;
    }
    if constexpr (RESTRICT) {
        load_restrict_block(y_first - (L + 2));
        tb_rc = tbn_rc;
        tb_rw = tbn_rw;
    }

    // register state: two-row history per level; F rows live in a ring of 8 (row t in slot t % 8: loaded at step
    // t - PF, read by level l at step t + l, by the residual at step t + S + 1); U rows in a ring of 2*PF.  The loop
    // body covers 8 steps, a multiple of every ring, so every slot index is a compile-time constant and nothing is
    // ever copied from slot to slot; a load never targets a register whose old value is still live.
    Row<COLS> older[L + 1], newer[L + 1];
#pragma unroll
    for (int l = 0; l <= L; ++l)
#pragma unroll
        for (int j = 0; j < COLS; ++j) older[l].v[j] = newer[l].v[j] = 0.0;
    constexpr int NU = PF < 4 ? 4 : 8;  // (a U row is consumed in the step it is due: PF < NU slots suffice)
    Row<COLS> fr[NFR], pu[NU];
    CoarseV<NCV> pc[NU];               // IN_PROLONG: coarse row (owner + 1) of the input row, 3 columns
    int q_own[NU];                // IN_PROLONG: owner coarse row of the input row
#pragma unroll
    for (int k = 0; k < NFR; ++k)
#pragma unroll
        for (int j = 0; j < COLS; ++j) fr[k].v[j] = 0.0;
#pragma unroll
    for (int k = 0; k < NU; ++k) {
#pragma unroll
        for (int j = 0; j < COLS; ++j) pu[k].v[j] = 0.0;
#pragma unroll
        for (int q = 0; q < NCV; ++q) pc[k].v[q] = 0.0;
        q_own[k] = -1;
    }

    if constexpr (LDSR) {  // (rows before the first arrival are read by pipeline levels nobody consumes: keep them finite)
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            s_g[wave][k][lane] = ring_t(real_t(0.0));
            s_f[wave][k][lane] = ringh_t(real_t(0.0));
        }
    }

    // wave-uniform row addresses, in bytes.  The load address follows the CLAMPED input row (every load is issued, see
    // load_row): it moves on only while the next row lies inside the window; the store address follows the unclamped
    // output row (stores are predicated).
    const char *in_b = reinterpret_cast<const char *>(a_in), *f_b = reinterpret_cast<const char *>(a_F);
    const int yc_first = y_first < av_lo ? av_lo : (y_first < av_hi ? y_first : av_hi - 1);
    unsigned long long ld_off = (unsigned long long)(yc_first - p.row_base) * row_bytes;  // row of the next load
    int ld_t = 0;                                                                          // its step index (row y_first + ld_t)
    long long st_off = (long long)(y_first - L - p.row_base) * (long long)row_bytes;       // row yin - L of the current step
    const unsigned inner_rows = (unsigned)(av_hi - av_lo - 1);

    // issue the loads of input row y_first + ld_t into their ring slots (u8 = ld_t % 8, compile-time at every call)
    auto fetch = [&](int u8) {
        if constexpr (IN != IN_ZERO && PRE == 0) pu[u8 % NU] = load_row<COLS, (MG_NT_LOADS >= 2)>(reinterpret_cast<const real_t *>(in_b + ld_off), col_off);
        fr[u8 % NFR] = load_row<COLS, (MG_NT_LOADS >= 1)>(reinterpret_cast<const real_t *>(f_b + ld_off), col_off);
        if constexpr (IN == IN_PROLONG) {
            // the UPPER coarse row of this input row travels with it, so every vector load of the loop is issued at
            // a fixed place PF steps before its use
            const int own_s = lane_value(tb_own, ld_t & 63);
            q_own[u8 % NU] = own_s;
            // local index of the coarse row (owner + 1), clamped to the window: rows fetched ahead of need may lie outside it
            // (never consumed).  Scalar arithmetic on the wave-uniform owner (a second table register cost two VGPRs)
            const int cr = own_s + 1 - p.coarse_base;
            const int crc = cr < 0 ? 0 : (cr < p.coarse_rows - 1 ? cr : p.coarse_rows - 1);
            pc[u8 % NU] = load_coarse_local<NCV>(a_coarse, p.Nc, crc, pc_col);
        }
        ++ld_t;
        // the clamped row moves on only inside the window: av_lo < y_first + ld_t < av_hi
        ld_off += ((unsigned)(y_first + ld_t - av_lo - 1) < inner_rows) ? row_bytes : 0u;
    };
#pragma unroll
    for (int k = 0; k < PF; ++k) fetch(k);

    if constexpr (IN == IN_PROLONG) {
        // before the first rotation hB must belong to the owner row of the first input row
        const int ys = yp_first > av_lo ? yp_first : av_lo;
        if (ys < av_hi && ys < y_end) {
            const int i0 = table_i(p.p_orow, ys);
            const CoarseV<NCV> c0 = load_coarse<NCV>(a_coarse, p.Nc, p.coarse_base, p.coarse_rows, i0, pc_col);
            interpolate(c0, hB);
            c_row = i0 - 1;
        }
    }

#ifdef MG_STREAM_TRACE
    const long long tr1 = wall_clock64();
#endif
    Row<COLS> gq[L + 1];   // LDS ring: dx2*F of the rows the levels of the coming step consume (zero before the first arrival)
#pragma unroll
    for (int l = 0; l <= L; ++l)
#pragma unroll
        for (int j = 0; j < COLS; ++j) gq[l].v[j] = 0.0;
    double acc = 0.0;
    const unsigned rows_own = (unsigned)(y1 - y0), rows_norm = (unsigned)(p.norm_y1 - p.norm_y0);
    // the loop body covers 8 rows, so the parity of the residual row is a compile-time property of the position in
    // the body once the parity of the chunk's first residual row is folded into the masks
    int nm_k0[COLS], nm_k1[COLS];  // masks of the residual rows at even / odd positions k
    const bool first_odd = ((y_first - L - 1) & 1) != 0;   // parity of the first residual row (wave-uniform)
    {
#pragma unroll
        for (int j = 0; j < COLS; ++j) {
            nm_k0[j] = first_odd ? nm_odd[j] : nm_even[j];
            nm_k1[j] = first_odd ? nm_even[j] : nm_odd[j];
        }
    }

    for (int t0 = 0; t0 < T; t0 += NB) {
        // table blocks: (t0 & 63) == 56 -> load the blocks that start 8 steps from now; == 0 -> they take over.
        // The owner block runs PF steps ahead of the others (it serves the prefetch), so it switches inside the body.
        if ((t0 & 63) == 64 - NB) {
            if constexpr (IN == IN_PROLONG && !TB_DIRECT) {
                load_own_block(yp_first + t0 + NB);
                load_weight_block(yp_first + t0 + NB);
            }
            if constexpr (RESTRICT) load_restrict_block(y_first + t0 + NB - (L + 2));
        }
        if ((t0 & 63) == 0 && t0 > 0) {
            if constexpr (IN == IN_PROLONG && TB_DIRECT) {
                load_weight_block(yp_first + t0);
            }
            if constexpr (IN == IN_PROLONG) {
                tb_yh = tbn_yh;
                tb_yl = tbn_yl;
            }
            if constexpr (RESTRICT) {
                tb_rc = tbn_rc;
                tb_rw = tbn_rw;
            }
        }
        // (the NB steps as calls with a compile-time k: `#pragma unroll` gives up on a body of this size at NB = 16, and a
        // rolled loop would index the rings at run time, i.e. move them to scratch memory)
        auto row_step = [&](auto k_tag) -> bool {
            constexpr int k = decltype(k_tag)::value;
            const int t = t0 + k;
            // (small levels: a chunk is a dozen rows, not a multiple of 8.  The LDS-ring form runs on big levels only, its
            // launcher makes a chunk's march a multiple of 8 steps, and steps past the end load clamped rows and store nothing)
            if (!LDSR && k > 0 && t >= T) return false;
            const int yin = y_first + t;
            Row<COLS> nw;
            if constexpr (IN == IN_ZERO || PRE > 0) {
#pragma unroll
                for (int j = 0; j < COLS; ++j) nw.v[j] = 0.0;
            } else {
                nw = pu[k % NU];
            }
            const int own_i = q_own[k % NU];
            if constexpr (IN == IN_PROLONG) {
                // the prefetch below reads the owner of row t + PF: from step 64 m - PF on that is the next block
                if (k == NB - PF && (t0 & 63) == 64 - NB) {
                    if constexpr (TB_DIRECT) load_own_block(yp_first + t0 + NB);
                    tb_own = tbn_own;
                }
                // the coarse row that came with this step's input is consumed AT ONCE (it feeds the horizontal interpolants
                // hA / hB only), so its registers are free again for the prefetch that follows
                if (own_i >= 0 && own_i != c_row) {  // the owner row advanced by one (host-checked): rotate
                    // (kept a wave-uniform BRANCH: if-converted, the rotation costs its selects and the interpolation
                    // on every row instead of on every other one)
                    asm volatile("" ::);
#pragma unroll
                    for (int j = 0; j < COLS; ++j) hA[j] = hB[j];
                    c_row = own_i;
                    interpolate(pc[k % NU], hB);  // row own_i + 1, loaded PF iterations ago
                }
            }
            fetch((k + PF) % NB);  // the row PF ahead, into the slots whose rows were retired (k + PF - 8 resp. k - PF)

            // U + P(coarse): doProlongation :700 as a gather, then doGridAddition :569 (U1 = U1 + U2), on the row the
            // prolongation belongs to -- the input row (PRE == 0) or level PRE's row yin - PRE.  own_i is wave-uniform.
            auto add_prolongation = [&](Row<COLS> &row) {
                if (own_i >= 0) {
                    const real_t own_yh = lane_value(tb_yh, t & 63), own_yl = lane_value(tb_yl, t & 63);
                    const real_t c_dx = p.c_dx, c_rcp = p.c_dx_rcp;
                    if constexpr (PACKED) {
#pragma unroll
                        for (int j = 0; j < COLS; j += 2) {
                            const pk_t yh2 = pk((float)own_yh, (float)own_yh), yl2 = pk((float)own_yl, (float)own_yl);
                            const pk_t num = pk((float)hA[j], (float)hA[j + 1]) * yh2 + pk((float)hB[j], (float)hB[j + 1]) * yl2;
                            const pk_t pv = pk_div_by_const(pk_div_by_const(num, (float)c_dx, (float)c_rcp), (float)c_dx, (float)c_rcp);
                            const pk_t sum = pk((float)row.v[j], (float)row.v[j + 1]) + pv;
                            row.v[j] = sum.x;
                            row.v[j + 1] = sum.y;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < COLS; ++j) {
                            const real_t num = hA[j] * own_yh + hB[j] * own_yl;
                            const real_t pv = div_by_const(div_by_const(num, c_dx, c_rcp), c_dx, c_rcp);
                            row.v[j] = row.v[j] + pv;
                        }
                    }
                }
            };
            if constexpr (IN == IN_PROLONG && PRE == 0) add_prolongation(nw);

            // levels 1..L: level l produces row yin-l from level l-1 rows yin-l-1, yin-l, yin-l+1
#pragma unroll
            for (int l = 1; l <= L; ++l) {
                const int y = yin - l;
                const int inner = ((unsigned)(y - 1) < (unsigned)(N - 2)) ? -1 : 0;  // 0 on the rim rows 0 and N-1
                const Row<COLS> c = newer[l - 1], so = older[l - 1];
                // dx2*F of row yin - l: from the LDS ring (formed once, when the row arrived), or formed here
                Row<COLS> g;
                if constexpr (LDSR) {
                    g = gq[l];   // requested at the end of the previous step
                } else {
                    const Row<COLS> &f = fr[(k - l + NB) % NB];
#pragma unroll
                    for (int j = 0; j < COLS; ++j) g.v[j] = dx2 * f.v[j];
                }
                const real_t west0 = from_lane_below(c.v[COLS - 1]);
                const real_t east_last = from_lane_above(c.v[0]);
                Row<COLS> o;
                if constexpr (PACKED) {
                    // the same expressions on the lane's column pairs (2P, 2P+1): the east neighbours of a pair are its second
                    // column and the first of the next pair (or of the next lane), the west neighbours the mirror image
#pragma unroll
                    for (int P = 0; P < COLS / 2; ++P) {
                        const int a = 2 * P, b = 2 * P + 1;
                        const pk_t q2 = pk((float)hi_bits_and(qc[a], inner), (float)hi_bits_and(qc[b], inner));
                        const pk_t g2 = pk((float)g.v[a], (float)g.v[b]);
                        pk_t o2;
                        if ((IN == IN_ZERO || PRE > 0) && MG_ZERO_LEVEL1 && l == 1) {
                            o2 = pk_fma(q2, -g2, pk(0.0f, 0.0f));   // (the first sweep from the zero field, see below)
                        } else {
                            const pk_t c2 = pk((float)c.v[a], (float)c.v[b]);
                            const pk_t e2 = pk((float)c.v[b], (float)(b == COLS - 1 ? east_last : c.v[b < COLS - 1 ? b + 1 : 0]));
                            const pk_t w2 = pk((float)(a == 0 ? west0 : c.v[a > 0 ? a - 1 : 0]), (float)c.v[a]);
                            pk_t t2 = pk((float)nw.v[a], (float)nw.v[b]) + pk((float)so.v[a], (float)so.v[b]);
                            t2 = t2 + e2;
                            t2 = t2 + w2;
                            t2 = pk_fma(pk(-4.0f, -4.0f), c2, t2) - g2;   // - 4U - dx2 F
                            o2 = pk_fma(q2, t2, c2);
                        }
                        o.v[a] = o2.x;
                        o.v[b] = o2.y;
                    }
                    older[l - 1] = c;
                    newer[l - 1] = nw;
                    nw = o;
                    if constexpr (IN == IN_PROLONG && PRE > 0) {
                        if (l == PRE) add_prolongation(nw);
                    }
                    continue;
                }
                if constexpr ((IN == IN_ZERO || PRE > 0) && MG_ZERO_LEVEL1) {
                    if (l == 1) {
                        // the first sweep from the zero field: every neighbour and the point itself are +0, so the sum,
                        // `- 4*U` and `U +` of the general expression below are exact no-ops -- the same bits without them
#pragma unroll
                        for (int j = 0; j < COLS; ++j) {
                            // q*(0 - dx2*F) + 0: `0 - p` is -p exactly, and whichever zero the product is, adding +0 gives +0:
                            // the negation rides on the fma's operand instead of costing a subtraction
                            o.v[j] = fused_mul_add(hi_bits_and(qc[j], inner), -g.v[j], real_t(0.0));
                        }
                        older[l - 1] = c;
                        newer[l - 1] = nw;
                        nw = o;
                        if constexpr (IN == IN_PROLONG && PRE > 0) {
                            if (l == PRE) add_prolongation(nw);
                        }
                        continue;
                    }
                }
#pragma unroll
                for (int j = 0; j < COLS; ++j) {
                    const real_t w = j == 0 ? west0 : c.v[j > 0 ? j - 1 : 0];
                    const real_t e = j == COLS - 1 ? east_last : c.v[j < COLS - 1 ? j + 1 : 0];
                    // src/MG_solver_CPU.cpp:590: U += 0.25*(U[i+1]+U[i-1]+U[j+1]+U[j-1] - 4U - dx^2 F)
                    // `- 4*U` through one fma: 4*U is exact, so fma(-4, U, a) is the same bits as a - 4*U (one VALU op less)
                    const real_t t4 = minus4(nw.v[j] + so.v[j] + e + w, c.v[j]) - g.v[j];
                    // `U + 0.25*t` through one fma as well: the product with a power of two is exact, so the fused form
                    // rounds once exactly like the sum does; with q = 0 on the rim (row or column) the point keeps its
                    // value, which replaces the rim selects
                    o.v[j] = fused_mul_add(hi_bits_and(qc[j], inner), t4, c.v[j]);
                }
                older[l - 1] = c;
                newer[l - 1] = nw;
                nw = o;
                if constexpr (IN == IN_PROLONG && PRE > 0) {
                    if (l == PRE) add_prolongation(nw);  // nw: the pre-smoothed row yin - PRE, as the `-1` node had it
                }
            }

            if constexpr (LDSR) {
                // the F row of THIS step (issued PF steps ago) goes to the rings; level 1 reads it in the next step.
                // Row yin has the parity of k + L + 1 (the march starts on an even row): an even row counts its even column
                const Row<COLS> &fin = fr[k % NFR];
                Row<COLS> gin;
#pragma unroll
                for (int j = 0; j < COLS; ++j) gin.v[j] = dx2 * fin.v[j];
                s_g[wave][k % NB][lane] = ring_pack<COLS>(gin);
                constexpr int par = (k + L + 1) & 1;
                if constexpr (COLS == 2) {
                    s_f[wave][k % NB][lane] = fin.v[par];
                } else {
                    ringh_t h;
                    h.x = fin.v[par];
                    h.y = fin.v[par + 2];
                    s_f[wave][k % NB][lane] = h;
                }
                // ... and the rows of dx2*F the L levels of the NEXT step consume are requested now (level 1's is the row just
                // written: the LDS operations of a wave execute in order), so their round trips lie under this step's tail
                // and the next step's head instead of in front of every level
#pragma unroll
                for (int l = 1; l <= L; ++l) {
                    gq[l] = ring_unpack<COLS>(s_g[wave][(k + 1 - l + 2 * NB) % NB][lane]);
                }
            }

            // nw is level L of row yin-L: the smoothed U
            {
                const int y = yin - L;
                if ((unsigned)(y - y0) < rows_own && lane_owns && !p.no_out) {
                    if (sizeof(real_t) != sizeof(double) && p.out_wide) {
                        // mixed precision: the last node of a cycle hands its result over in fp64 (exact
                        // widening) instead of leaving it to a separate conversion pass
                        double *w = reinterpret_cast<double *>(reinterpret_cast<char *>(p.out_wide) + 2 * st_off + 2u * col_st);  // 8 B per element here
                        typedef double wide2_t __attribute__((ext_vector_type(2)));
                        if constexpr (COLS % 2 == 0) {  // 16-byte stores (column pairs are 16-byte aligned in the fp64 array)
#pragma unroll
                            for (int j = 0; j < COLS; j += 2) {
                                wide2_t t;
                                t.x = (double)nw.v[j];
                                t.y = (double)nw.v[j + 1];
                                // (4 columns per lane: the two 16-byte halves of a lane's 32 bytes come from two instructions, each
                                // strided by 32 B; a streaming store would write the half lines out one by one: let L2 merge them)
                                if (COLS == 2) __builtin_nontemporal_store(t, reinterpret_cast<wide2_t *>(w + j));
                                else *reinterpret_cast<wide2_t *>(w + j) = t;
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < COLS; ++j) __builtin_nontemporal_store((double)nw.v[j], w + j);
                        }
                    } else {
                        store_row<COLS, NT>(reinterpret_cast<real_t *>(reinterpret_cast<char *>(a_out) + st_off), col_st, nw);
                    }
                }
            }

            // residual stage, row yin-S-1 (src/MG_solver_CPU.cpp:560 and the error sums :611)
            if (want_res) {
                const int y = yin - L - 1;
                const bool mine_row = (unsigned)(y - y0) < rows_own;  // each point counted once
                const int inner = ((unsigned)(y - 1) < (unsigned)(N - 2)) ? -1 : 0;
                const int cm = (mine_row && (unsigned)(y - p.norm_y0) < rows_norm) ? inner : 0;  // -1: the row counts
                const Row<COLS> c = newer[L], so = older[L];
                if constexpr (HALF) {
                    // norm only: row y has the parity of k, its counted columns are j = k & 1 (+ 2, + 4 ...: xl is even)
                    const ringh_t fh = s_f[wave][(k - L - 1 + 2 * NB) % NB][lane];
#pragma unroll
                    for (int j = k & 1; j < COLS; j += 2) {
                        real_t fv;
                        if constexpr (COLS == 2) fv = fh;
                        else fv = fh[j >> 1];
                        const real_t w = j == 0 ? from_lane_below(c.v[COLS - 1]) : c.v[j > 0 ? j - 1 : 0];
                        const real_t e = j == COLS - 1 ? from_lane_above(c.v[0]) : c.v[j < COLS - 1 ? j + 1 : 0];
                        const real_t r = inv * minus4(nw.v[j] + so.v[j] + e + w, c.v[j]) - fv;
                        const int am = ((lane_owns && !col_edge[j]) ? -1 : 0) & cm;
                        acc += fabs(bits_and((double)r, am));
                    }
                    older[L] = newer[L];
                    newer[L] = nw;
                    st_off += row_bytes;
                    return true;
                }
                const Row<COLS> &f = fr[(k - L - 1 + 2 * NB) % NFR];
                const real_t west0 = from_lane_below(c.v[COLS - 1]);
                const real_t east_last = from_lane_above(c.v[0]);
                Row<COLS> d;
#pragma unroll
                for (int j = 0; j < COLS; ++j) {
                    const real_t w = j == 0 ? west0 : c.v[j > 0 ? j - 1 : 0];
                    const real_t e = j == COLS - 1 ? east_last : c.v[j < COLS - 1 ? j + 1 : 0];
                    const real_t r = inv * minus4(nw.v[j] + so.v[j] + e + w, c.v[j]) - f.v[j];
                    if constexpr (RESTRICT) {
                        d.v[j] = r * hi_bits_and(ms[j], inner);  // sign flip :277-280 and the zero rim in one exact product
                    } else {
                        d.v[j] = r;  // (masked and signed where it is stored, below: most launches store no residual)
                    }
                    // (row+col) even interior points only, :610/:617; norms are accumulated in fp64 whatever the field type
                    if constexpr (EVEN_START) {
                        // row y has the parity of k: only the columns of that parity can count (the others added +0.0)
                        if ((j & 1) == (k & 1)) acc += fabs(bits_and((double)r, ((lane_owns && !col_edge[j]) ? -1 : 0) & cm));
                    } else {
                        const int am = ((k & 1) ? nm_k1[j] : nm_k0[j]) & cm;
                        acc += fabs(bits_and((double)r, am));
                    }
                }
                if constexpr (!RESTRICT) {
                    if (mine_row && lane_owns && p.D) {
                        Row<COLS> ds;
#pragma unroll
                        for (int j = 0; j < COLS; ++j) {
                            const real_t dv = (inner != 0 && !col_edge[j]) ? d.v[j] : real_t(0.0);
                            ds.v[j] = p.d_sign < 0 ? -dv : dv;  // the driver's sign flip :277-280
                        }
                        store_row<COLS, false>(reinterpret_cast<real_t *>(reinterpret_cast<char *>(p.D) + (st_off - (long long)row_bytes)), col_st, ds);
                    }
                } else {
                    if (mine_row && lane_owns && p.D)
                        store_row<COLS, false>(reinterpret_cast<real_t *>(reinterpret_cast<char *>(p.D) + (st_off - (long long)row_bytes)), col_st, d);
                }

                if constexpr (RESTRICT) {
                    // doRestriction :656-678 on rows (y-1, y) of the signed residual: coarse row
                    // rc has its lower-left sample in fine row y-1
                    // (rc_row / rc_w: wave-uniform, -1 = no coarse row has its lower-left sample in fine row y-1 of this chunk)
                    const int rc_row = lane_value(tb_rc, t & 63);
                    if (rc_row >= 0) {
                        const real_t wc = lane_value(tb_rw, t & 63), wd = real_t(1.0) - wc;  // c, d of :664-666
                        const real_t p_up = from_lane_above(d_prev.v[0]);
                        const real_t q_up = from_lane_above(d.v[0]);
                        real_t *crow = a_Fc + (size_t)(rc_row - p.fc_base) * p.M;
#pragma unroll
                        for (int q = 0; q < NS; ++q) {
                            // the column after the pair: the lane's next pair, or the next lane's first column
                            const real_t p_next = q + 1 < NS ? d_prev.v[q + 1 < NS ? 2 * q + 2 : 0] : p_up;
                            const real_t q_next = q + 1 < NS ? d.v[q + 1 < NS ? 2 * q + 2 : 0] : q_up;
                            const real_t u0 = rc_shift[q] ? d_prev.v[2 * q + 1] : d_prev.v[2 * q];
                            const real_t u1 = rc_shift[q] ? p_next : d_prev.v[2 * q + 1];
                            const real_t u2 = rc_shift[q] ? d.v[2 * q + 1] : d.v[2 * q];
                            const real_t u3 = rc_shift[q] ? q_next : d.v[2 * q + 1];
                            // :676  U_c = b*d*U_f[f] + a*d*U_f[f+1] + c*b*U_f[f+N] + a*c*U_f[f+N+1]
                            const real_t vc = rw_b[q] * wd * u0 + rw_a[q] * wd * u1 + wc * rw_b[q] * u2 + rw_a[q] * wc * u3;
                            if (rc_col[q] >= 0) crow[rc_col[q]] = vc;
                        }
                        if (first_col_lane) crow[0] = 0.0;
                        if (last_col_lane) crow[p.M - 1] = 0.0;
                    }
                    d_prev = d;
                }
            }
            older[L] = newer[L];
            newer[L] = nw;
            st_off += row_bytes;
            return true;
        };
        unrolled_while(row_step, std::make_integer_sequence<int, NB>{});
    }
#ifdef MG_STREAM_TRACE
    if (p.trace && (threadIdx.x & 63) == 0) {  // per wave: start, end of prologue, end of loop (100 MHz ticks); [0..3]: legacy record of the middle tile
        const long long tr2 = wall_clock64();
        long long *rec = p.trace + 8 + 4 * ((size_t)tile * WAVES_PER_WG + wave);
        rec[0] = tr0;
        rec[1] = tr1;
        rec[2] = tr2;
        rec[3] = (long long)blockIdx.x;
        if (tile == p.n_blocks / 2 && wave == 0) {
            p.trace[0] = tr0;
            p.trace[1] = tr1;
            p.trace[2] = tr2;
            p.trace[3] = T;
        }
    }
#endif
    if (a_part) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
        if (lane == 0) a_part[(size_t)tile * WAVES_PER_WG + wave] = acc;
    }
}

#ifndef MG_STREAM_KERNEL_ONLY   // (register experiments compile single instantiations of the kernel without the launchers)
// One launch: tile the grid for ONE resident round of workgroups (measured occupancy of
// this instantiation x CUs) where the grid is large enough, never fewer than 8 rows per
// chunk (each chunk re-reads 2(S+1) halo rows), then the fixed-order error reduction.
template <int S, int COLS, int IN, bool RESTRICT, int PF = PF_DEFAULT, bool NT = false, int PRE = 0>
void launch_k(hipStream_t s, StreamParams p, double *err_out)
{
    static int blocks_per_cu = 0;
    if (blocks_per_cu == 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_jacobi_stream<S, COLS, IN, RESTRICT, PF, NT, PRE>, 64 * WAVES_PER_WG, 0) != hipSuccess || n < 1) {
            (void)hipGetLastError();
            n = 2;
        }
        blocks_per_cu = n > 8 ? 8 : n;
    }
    const int N = p.N;
    const int own = p.own_y1 - p.own_y0;  // rows this launch updates (N on a single GPU)
    if (own <= 0) return;
    constexpr int OW = 64 * COLS - 2 * Halo<S + PRE, RESTRICT, COLS>::value;
    const int strips = (N + OW - 1) / OW;
    const int groups = (strips + WAVES_PER_WG - 1) / WAVES_PER_WG;
    static const int resident_pct = [] { const char *e = getenv("MG_RESIDENT_PCT"); return e ? atoi(e) : 100; }();
    const int resident = ctx().n_cu * blocks_per_cu * resident_pct / 100;
    const int nb = p.batch ? p.n_batch : 1;   // the instances of a batch share the resident round
    int chunks = resident / (groups * nb);
    // small grids are latency bound on the length of a wave's march: shorter chunks, more waves
    static const int min_rows_forced = [] { const char *e = getenv("MG_MIN_ROWS"); return e ? atoi(e) : 0; }();
    const int min_rows = min_rows_forced ? min_rows_forced : (N <= 256 ? 2 : N <= 1024 ? 4 : 8);
    const int max_chunks = (own + min_rows - 1) / min_rows;
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks < 1) chunks = 1;
    int rows = (own + chunks - 1) / chunks;
    static const int rows_cap = [] { const char *e = getenv("MG_MAX_ROWS"); return e ? atoi(e) : 0; }();
    if (rows_cap > 0 && rows > rows_cap) rows = rows_cap;  // tuning knob: more, shorter tiles
    if (LdsRing<COLS, PRE>::value) {
        // even (the norm-only residual stage starts every chunk on an even row) and a march of rows + 2 (L + 1) steps that
        // is a multiple of the 8-step loop body (which has no exit inside)
        const int march = rows + 2 * (S + PRE + 1);
        rows += (8 - march % 8) % 8;
    } else if (COLS >= 2) {
        rows += rows & 1;   // chunks start on even rows (the kernel would otherwise march one row more per chunk)
    }
    chunks = (own + rows - 1) / rows;
    p.rows_per_chunk = rows;
    p.groups = groups;
    p.n_blocks = chunks * groups;
    p.part = nullptr;
#ifdef MG_STREAM_TRACE
    static long long *trace_dev = nullptr;
    if (!trace_dev) (void)hipMalloc((void **)&trace_dev, (8 + 4 * 4 * 4096) * sizeof(long long));
    (void)hipMemsetAsync(trace_dev, 0, (8 + 4 * 4 * 4096) * sizeof(long long), s);
    p.trace = trace_dev;
#endif
    const size_t n_part = (size_t)p.n_blocks * WAVES_PER_WG;
    const bool want_norm = err_out || (p.batch && p.err_outs);
    if (want_norm) {
        p.part = norm_partials(n_part * (size_t)nb);  // every wave of every tile writes its slot
        if (!p.part) return;
    }
    p.part_stride = (int)n_part;
    const int grid = ((p.n_blocks + 7) / 8) * 8;   // (a multiple of 8: with x fastest in the dispatch order, x & 7 stays the XCD of a workgroup for every y)
    hipLaunchKernelGGL((k_jacobi_stream<S, COLS, IN, RESTRICT, PF, NT, PRE>), dim3(grid, nb), dim3(64 * WAVES_PER_WG), 0, s, p);
#ifdef MG_STREAM_TRACE
    {
        long long t[4];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(t, trace_dev, sizeof t, hipMemcpyDeviceToHost);
        fprintf(stderr, "[stream trace] N=%d IN=%d R=%d tiles=%d rows/chunk=%d: prologue %.2f us, %lld steps %.2f us (%.3f us/step)\n", N, IN, (int)RESTRICT,
                p.n_blocks, rows, (t[1] - t[0]) * 0.01, t[3], (t[2] - t[1]) * 0.01, (t[2] - t[1]) * 0.01 / (double)t[3]);
        if (N >= 4096 && p.n_blocks <= 4096) {  // spread of the waves' lifetimes
            const size_t nw = (size_t)p.n_blocks * WAVES_PER_WG;
            std::vector<long long> all(4 * nw);
            (void)hipMemcpy(all.data(), trace_dev + 8, all.size() * sizeof(long long), hipMemcpyDeviceToHost);
            long long t_min = 0x7fffffffffffffffll, t_max = 0;
            std::vector<double> start, life, endt;
            for (size_t w = 0; w < nw; ++w) {
                if (all[4 * w] == 0) continue;  // a wave past the grid's last strip left no record
                if (all[4 * w] < t_min) t_min = all[4 * w];
                if (all[4 * w + 2] > t_max) t_max = all[4 * w + 2];
            }
            double xcd_end[8] = {0}, xcd_start[8] = {0};
            for (size_t w = 0; w < nw; ++w) {
                if (all[4 * w] == 0) continue;
                start.push_back((all[4 * w] - t_min) * 0.01);
                endt.push_back((all[4 * w + 2] - t_min) * 0.01);
                life.push_back((all[4 * w + 2] - all[4 * w]) * 0.01);
                const int x = (int)(all[4 * w + 3] & 7);
                if (endt.back() > xcd_end[x]) xcd_end[x] = endt.back();
                if (start.back() > xcd_start[x]) xcd_start[x] = start.back();
            }
            auto pct = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
            fprintf(stderr, "   waves %zu: start p50 %.1f p99 %.1f max %.1f us | lifetime p1 %.1f p50 %.1f p99 %.1f max %.1f | end p1 %.1f p50 %.1f p99 %.1f max %.1f | span %.1f us\n",
                    nw, pct(start, .5), pct(start, .99), pct(start, 1.0), pct(life, .01), pct(life, .5), pct(life, .99), pct(life, 1.0), pct(endt, .01),
                    pct(endt, .5), pct(endt, .99), pct(endt, 1.0), (t_max - t_min) * 0.01);
            fprintf(stderr, "   last wave per XCD ends at:");
            for (int x = 0; x < 8; ++x) fprintf(stderr, " %.1f", xcd_end[x]);
            fprintf(stderr, " us\n");
        }
    }
#endif
    // a slab launch leaves its RAW partial sum; the caller combines the slabs in rank order
    if (p.batch && p.err_outs) {
        for (int i = 0; i < nb; ++i)
            if (p.err_outs[i]) norm_finish(s, p.part + (size_t)i * n_part, n_part, p.raw_norm ? -1 : N, p.err_outs[i]);
    } else if (err_out) {
        norm_finish(s, p.part, n_part, p.raw_norm ? -1 : N, err_out);
    }
}

template <int S, int PF>
void launch_variant(hipStream_t s, const StreamParams &p, double *err_out)
{
    const bool restrict_out = p.Fc != nullptr, prolong_in = p.coarse != nullptr, zero = p.in == nullptr;
    // non-temporal stores of U: the fused `1` node of the biggest levels only (measured with real nt stores, three
    // alternating runs per build in one session: 358 -> 347 us at N = 8192; at 4096 a loss, 109 -> 118 us: the next
    // launch reads that U as its coarse input; the `-1` node loses 4 % with them: MG_NT_MIN_N, default 8192)
    const bool nt = p.N >= p.nt_min_n;
    if constexpr (sizeof(real_t) == 4) {
        // fp32 fields, 16 B per lane: the same kernel with 4 columns per lane (8 B per lane leaves it bound by its
        // instruction stream at half the bytes per instruction of the fp64 build).  Not the recomputing `1` node: its
        // 6-level pipeline needs 267 VGPRs with 4 columns (one wave per SIMD: 270 us at N = 8192) and 144 with 2
        // (Not the recomputing `1` node: measured at N = 8192 with its F rows in LDS (LdsRing<4, PRE>: 201 VGPRs, two waves per
        // SIMD) the 4-column form takes 237 us with scalar and 233 us with packed arithmetic against 212 us of the 2-column
        // form at four waves per SIMD -- its instruction stream carries ~155 scalar instructions per row step, SGPR spills
        // included; profiles/r04_f32_packed_experiment.txt.  The instantiation is not built.)
        if (p.cols4 && !(prolong_in && p.pre)) {
            if (restrict_out) {
                if (zero) launch_k<S, 4, IN_ZERO, true, PF>(s, p, err_out);
                else launch_k<S, 4, IN_LOAD, true, PF>(s, p, err_out);
            } else if (prolong_in) {
                if (nt) launch_k<S, 4, IN_PROLONG, false, PF, true>(s, p, err_out);
                else launch_k<S, 4, IN_PROLONG, false, PF>(s, p, err_out);
            } else {
                if (zero) launch_k<S, 4, IN_ZERO, false, PF>(s, p, err_out);
                else launch_k<S, 4, IN_LOAD, false, PF>(s, p, err_out);
            }
            return;
        }
    }
    if (p.N % 2 != 0) {  // 8 B lanes: odd row pitch; the fused transfer stages are not built for it
        if (zero) launch_k<S, 1, IN_ZERO, false, PF>(s, p, err_out);
        else launch_k<S, 1, IN_LOAD, false, PF>(s, p, err_out);
    } else if (restrict_out) {
        if (zero) launch_k<S, 2, IN_ZERO, true, (PF == 2 && S <= 3 ? MG_PF_DOWN : PF)>(s, p, err_out);
        else launch_k<S, 2, IN_LOAD, true, PF>(s, p, err_out);
    } else if (prolong_in) {
        if (p.pre != 0) {  // the pre-smoothed U recomputed, not read (the caller checked recompute_available)
            bool done = false;
            auto with_pre = [&](auto pre_tag) {
                constexpr int PRE = decltype(pre_tag)::value;
                if constexpr (recompute_instantiated(PRE, S) && PF == 2) {
                    if (p.pre == PRE && !done) {
                        // (rows in flight: with the F ring in LDS the registers for a deeper prefetch are there, MG_PF_UP)
                        constexpr int PFU = LdsRing<2, PRE>::value ? MG_PF_UP : PF;
                        if (nt) launch_k<S, 2, IN_PROLONG, false, PFU, true, PRE>(s, p, err_out);
                        else launch_k<S, 2, IN_PROLONG, false, PFU, false, PRE>(s, p, err_out);
                        done = true;
                    }
                }
            };
            with_pre(std::integral_constant<int, 1>{});
            with_pre(std::integral_constant<int, 2>{});
            with_pre(std::integral_constant<int, 3>{});
            with_pre(std::integral_constant<int, 4>{});
            // (no instantiation for this pair / prefetch depth: never read the absent input through the plain node)
            if (!done) fail(MG_ERR_UNSUPPORTED, "jacobi_stream: no recomputing `1` node for %d + %d sweeps, prefetch depth %d", p.pre, S, PF);
            return;
        }
        if (nt) launch_k<S, 2, IN_PROLONG, false, PF, true>(s, p, err_out);
        else launch_k<S, 2, IN_PROLONG, false, PF>(s, p, err_out);
    } else {
        if (zero) launch_k<S, 2, IN_ZERO, false, PF>(s, p, err_out);
        else launch_k<S, 2, IN_LOAD, false, PF>(s, p, err_out);
    }
}

template <int S>
void launch_steps(hipStream_t s, const StreamParams &p, double *err_out)
{
    // rows in flight per lane: PF = 2 (a FIFO of 4 slots).  A deeper FIFO (PF = 3 at the same occupancy, PF = 4, 8) costs registers
    // and a longer prologue and measured slower at every size from 128 to 8192 once all loads were
    // unconditional (fused prolongation: 18.1 vs 21.1 us at N = 1024, 10.9 vs 13.7 us at N = 128).  Round 2, for the `-1` node
    // that no longer stores U (it loads F alone): PF = 3 gives 147 against 151 us at N = 8192, nothing below -- not taken.
    if constexpr (S <= 3 && MG_PF == 3) launch_variant<S, 3>(s, p, err_out);
    else launch_variant<S, 2>(s, p, err_out);
}


// entry point of this instantiation: same contract as k::jacobi_stream (mg_internal.h) in the
// field type real_t; tb carries the tables the fused stages need
inline void run(hipStream_t s, int N, real_t dx2, real_t inv, const real_t *in, const real_t *F, real_t *out, int steps,
                double *err_out, real_t *D_out, int d_sign, const real_t *coarse, int Nc, real_t *Fc, int M,
                const StreamTables &tb, const RowWindow *fine_w, const RowWindow *coarse_w, const RowWindow *fc_w,
                double *out_wide = nullptr, int pre = 0, bool no_out = false, const NodeBatch *batch = nullptr)
{
    if (batch && (fine_w || coarse_w || fc_w || D_out || out_wide || batch->n < 1)) {
        fail(MG_ERR_ARG, "jacobi_stream: a batch of instances runs whole grids without a stored residual");
        return;
    }
    if (pre != 0 && !(recompute_instantiated(pre, steps) && coarse && !Fc)) {
        fail(MG_ERR_ARG, "jacobi_stream: recomputed pre-smoothing exists for pre + steps <= 6 sweeps of the fused `1` node (pre=%d steps=%d)", pre, steps);
        return;
    }
    if (pre != 0 && D_out) {
        fail(MG_ERR_ARG, "jacobi_stream: the recomputing `1` node forms the error norm only, it stores no residual");
        return;
    }
    if (steps < 1 || steps > MAX_S) {
        fail(MG_ERR_ARG, "jacobi_stream: %d sweeps per launch (1..%d)", steps, MAX_S);
        return;
    }
    if ((coarse || Fc) && (N < 4 || N % 2 != 0)) {
        fail(MG_ERR_ARG, "jacobi_stream: fused transfer stages need an even grid size (N=%d)", N);
        return;
    }
    StreamParams p = {};
    if (batch) {
        p.batch = batch->dev;
        p.n_batch = batch->n;
        p.err_outs = batch->err_outs;
    }
    p.N = N;
    p.dx2 = dx2;
    p.inv = inv;
    p.in = in;
    p.F = F;
    p.out = out;
    p.out_wide = out_wide;
    p.pre = pre;
    p.no_out = no_out ? 1 : 0;
    p.D = D_out;
    p.d_sign = d_sign;
    p.row_base = fine_w ? fine_w->base : 0;
    p.rows_local = fine_w ? fine_w->rows : N;
    p.own_y0 = fine_w ? fine_w->own_lo : 0;
    p.own_y1 = fine_w ? fine_w->own_hi : N;
    p.norm_y0 = fine_w && fine_w->norm_lo >= 0 ? fine_w->norm_lo : p.own_y0;
    p.norm_y1 = fine_w && fine_w->norm_lo >= 0 ? fine_w->norm_hi : p.own_y1;
    p.raw_norm = fine_w ? 1 : 0;
    static const int cols4_min = [] { const char *e = getenv("MG_F32_COLS4_MIN_N"); return e ? atoi(e) : 8192; }();  // (measured: a gain only where the launch is bandwidth-bound, N >= 8192)
    p.cols4 = (sizeof(real_t) == 4 && N % 4 == 0 && N >= cols4_min && (!coarse || tb.p_cols4_ok)) ? 1 : 0;
    static const int nt_min = [] { const char *e = getenv("MG_NT_MIN_N"); return e ? atoi(e) : 8192; }();
    p.nt_min_n = nt_min;
    if (coarse) {
        p.coarse_base = coarse_w ? coarse_w->base : 0;
        p.coarse_rows = coarse_w ? coarse_w->rows : Nc;
        p.coarse = coarse;
        p.Nc = Nc;
        p.p_orow = tb.p_orow;
        p.p_ocol = tb.p_ocol;
        p.p_rhi = tb.p_rhi;
        p.p_rlo = tb.p_rlo;
        p.p_chi = tb.p_chi;
        p.p_clo = tb.p_clo;
        p.c_dx = tb.c_dx;
        p.c_dx_rcp = tb.c_dx_rcp;
    }
    if (Fc) {
        p.fc_base = fc_w ? fc_w->base : 0;
        p.fc_rows = fc_w ? fc_w->rows : M;
        p.Fc = Fc;
        p.M = M;
        p.r_inv = tb.r_inv;
        p.r_w = tb.r_w;
        p.r_wf = tb.r_wf;
    }
    switch (steps) {
        case 1: launch_steps<1>(s, p, err_out); break;
        case 2: launch_steps<2>(s, p, err_out); break;
        case 3: launch_steps<3>(s, p, err_out); break;
        default: launch_steps<4>(s, p, err_out); break;
    }
}

#endif  // MG_STREAM_KERNEL_ONLY

}  // namespace MG_REAL_NS
}  // namespace k
}  // namespace mg
