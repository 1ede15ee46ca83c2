// mg_internal.h -- engine-internal declarations shared by the host files and the
// kernel file.  Nothing here is part of the C ABI (include/mg_hip.h).
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "mg_hip.h"

namespace mg {

// ---------------------------------------------------------------------------
// errors: the reference printf+exit(1)s (src/MG_solver_GPU.cu:58-62,1286-1289)
// ---------------------------------------------------------------------------
enum ErrorCode {
    MG_OK = 0,
    MG_ERR_HIP = 1,
    MG_ERR_ARG = 2,
    MG_ERR_UNSUPPORTED = 3,
    MG_ERR_NOT_INIT = 4,
    MG_ERR_CYCLE_FILE = 5,
    MG_ERR_COMM = 6,
};

void fail(int code, const char *fmt, ...);
bool hip_ok(hipError_t e, const char *what, const char *file, int line);
#define MG_HIP(expr) ::mg::hip_ok((expr), #expr, __FILE__, __LINE__)

// ---------------------------------------------------------------------------
// device tables for restriction / prolongation, cached per (N, M)
// ---------------------------------------------------------------------------
struct RestrictTable {
    int *lo = nullptr;     // [M] lower-left fine index
    double *w = nullptr;   // [M] weight a (resp. c)
    int *inv = nullptr;    // [N] fine index -> interior coarse index with lo[] == it, else -1
    double *inv_w = nullptr; // [N] w[inv[x]] (0 where inv[x] < 0)
    float *w_f = nullptr, *inv_w_f = nullptr;  // the same weights rounded to fp32 (mixed-precision mode)
    bool fusable = false;  // lo[] strictly increasing by >= 2: one coarse sample per column pair
};
struct ProlongTable {
    int *owner_row = nullptr, *owner_col = nullptr;        // [M]
    double *row_hi = nullptr, *row_lo = nullptr;           // [M] (c3y - f_y), (f_y - c1y)
    double *col_hi = nullptr, *col_lo = nullptr;           // [M] (c2x - f_x), (f_x - c1x)
    float *row_hi_f = nullptr, *row_lo_f = nullptr, *col_hi_f = nullptr, *col_lo_f = nullptr;  // rounded to fp32
    double c_dx = 0.0;
    bool fusable = false;  // every fine index owned, owners advance by <= 1 per fine index
    bool fusable4 = false; // ... and 4 aligned fine columns span at most 3 coarse cells (4 columns per lane, fp32)
    // owner_row[k] == owner_col[k] == min(k*(N-1)/(M-1), N-2) in integer arithmetic for every fine index k (checked entry
    // by entry against the tables built from the reference's ceil() expressions; M <= 4096): a kernel may then form the
    // owner -- an ADDRESS ingredient -- itself instead of waiting for a table load before it can issue its coarse loads
    bool closed_form = false;
};

// a window of grid rows held in a local array: rows [base, base+rows) of the global grid
// (base may be negative / extend past N: those rows are never touched), of which this rank
// owns [own_lo, own_hi).  1-D row-slab decomposition, BASELINE.json north_star.
struct RowWindow {
    int base = 0, rows = 0, own_lo = 0, own_hi = 0;
    // rows that count towards the error norm; -1: the rows [own_lo, own_hi).  A slab launch may
    // update more rows than its rank owns (redundant halo rows instead of a ghost exchange).
    int norm_lo = -1, norm_hi = -1;
};

// ---------------------------------------------------------------------------
// caching device pool (replaces malloc/free of the level arrays)
// ---------------------------------------------------------------------------
class Pool {
public:
    void *get(size_t bytes);
    void put(void *p);
    void trim();
    size_t bytes_held() const { return held_; }
    // Reuse of a returned block relies on stream order (the next user enqueues behind the last one).  While a cycle plan
    // traces its node program for a batched schedule (mg_cycle.cpp: build_schedule) every visit of a level must keep
    // arrays of its own: with park(true) returned blocks are set aside instead of becoming available, until
    // release_parked() -- called when the schedule is dropped.
    void park(bool on) { park_ = on; }
    void release_parked();
private:
    std::multimap<size_t, void *> free_;   // size -> block
    std::map<void *, size_t> live_;        // block -> size
    std::vector<std::pair<size_t, void *>> parked_;
    bool park_ = false;
    size_t held_ = 0;
};

// STREAM_ONLY: the streaming kernel also for a bare single sweep of a large grid (which STREAM hands to the
// one-row-per-block pair kernel): lets the bench report the S = 1 streaming kernel's own rate
enum Smoother { SMOOTHER_STREAM = 0, SMOOTHER_SIMPLE = 1, SMOOTHER_STREAM_ONLY = 2 };

struct NodeOp;   // (below: one fused node launch as the cycle driver's dataflow trace records it)

struct Context {
    bool ready = false;
    int device = -1;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    int n_cu = 256;
    Pool pool;
    Pool *active_pool = nullptr;           // plan-private arena while a cycle plan executes
    int recompute_min_override = 0;        // > 0 while a batched schedule is traced: the recomputing node pair from this N on
    std::vector<NodeOp> *trace = nullptr;  // != nullptr: fused nodes are recorded, not launched (mg_cycle.cpp: build_schedule)
    bool trace_failed = false;             // ... and one of them was not a single fused launch
    Smoother smoother = SMOOTHER_STREAM;
    int source_mode = 0;                   // mg_set_source: 0 auto (device when it reproduces the host's libm bit for bit), 1 host, 2 device
    int source_identical = -1;             // result of the self-check: -1 not run yet, 0 differs, 1 identical
    // reduction scratch: per-block partial sums + scalar slots
    double *partials = nullptr;
    size_t partials_cap = 0;
    std::vector<void *> retired;          // outgrown scratch, kept alive for captured graphs
    // deferred norm reductions: while a cycle plan runs, the smoothing kernels only leave
    // per-wave partial sums in an arena; ONE kernel per flush finishes all of them
    struct PendingNorm { const double *part; int n; int N; double *out; };
    bool defer_norms = false;
    bool norms_at_window_end = false;      // never flush in mid-window (nodes may still be running on other streams)
    std::vector<PendingNorm> pending_norms;
    double *norm_arena = nullptr;
    size_t norm_arena_cap = 0, norm_arena_used = 0;
    size_t norm_window_total = 0;          // partials requested since the last flush (sizes the arena for the next window)
    double *scalars = nullptr;            // [64] device scalars (errors, norms)
    int *gs_state = nullptr;              // [4]: done flag, iteration count, ...
    double *host_scalars = nullptr;       // pinned [64]
    int *host_ints = nullptr;             // pinned [4]
    std::map<std::pair<int, int>, RestrictTable> rtab;
    std::map<std::pair<int, int>, ProlongTable> ptab;
    int last_error = 0;
    std::string last_error_text;
    bool abort_on_error = true;
    // profiling (mg_profile_begin/end)
    bool profiling = false;
    int profile_min_N = 0;
    int profile_every = 1;                 // mg_profile_sample: time the launches of every k-th cycle window only
    long profile_window = 0;               // windows enqueued since mg_profile_begin
    struct ProfRec { std::string name; int N; double bytes; hipEvent_t e0, e1; };
    std::vector<ProfRec> prof;
    std::vector<hipEvent_t> event_pool;
};

// live kernel timing: RAII event pair around a launch (no-op unless profiling is on)
struct ProfScope {
    ProfScope(const char *name, int N, double algo_bytes, hipStream_t stream = nullptr);  // nullptr: the engine's stream
    ~ProfScope();
    int slot = -1;
    hipStream_t on = nullptr;
};

Context &ctx();
bool require_ready(const char *who);
double *partials(size_t n);   // device scratch for at least n doubles
Pool &scratch_pool();         // where operator-internal scratch comes from
// partial-sum storage + registration of a doSmoothing error reduction (deferred while a plan
// runs, finished immediately otherwise).  flush_norms() finishes everything pending.
double *norm_partials(size_t n);
void norm_finish(hipStream_t s, const double *part, size_t n, int N, double *out);
void flush_norms();

const RestrictTable &restrict_table(int N, int M);
const ProlongTable &prolong_table(int N, int M);
// host-side builders (exact reference expressions)
void build_restriction_table(int N, int M, int *lo, double *w);
void build_prolongation_table(int N, int M, int axis, int *owner, double *w_hi, double *w_lo);
bool restriction_table_in_bounds(int N, int M, const int *lo);

// getSource (src/MG_solver_CPU.cpp:468-493) for grid rows [row_lo, row_hi) evaluated on the host
// (libm exp) into dev_dst, which points at row row_lo
void fill_source_rows(int N, double L, double min_x, double min_y, int row_lo, int row_hi, double *dev_dst);
// smoothing on a row window with optional fused transfer stages (mg_abi.cpp)
struct SlabFusion {
    const double *coarse = nullptr;
    int Nc = 0;
    double *Fc = nullptr;
    int M = 0;
    RowWindow fine_w, coarse_w, fc_w;
    int pre = 0;          // `1` launch: the pre-smoothed field is recomputed (pre sweeps from zero), U_in is not read
    bool no_out = false;  // `-1` launch: the smoothed field is not stored
    double *out_wide = nullptr;  // fp32 fields: the result goes to this fp64 window (same geometry, exact widening) instead of U_out
};
void slab_smooth(int N, double L, const double *U_in, double *U_out, const double *F, int step, double *raw_norm_out,
                 const SlabFusion &sf);
// the same launch on fp32 fields (mixed-precision slabs): the pointers inside sf are float arrays
void slab_smooth_f32(int N, double L, const float *U_in, float *U_out, const float *F, int step, double *raw_norm_out,
                     const SlabFusion &sf);

// RCCL transport (mg_comm.cpp)
bool comm_ready();
void comm_set_stream(hipStream_t s);   // stream of the following operations (nullptr: the engine's stream)
int comm_rank();
int comm_size();
void comm_group_begin();
void comm_group_end();
void comm_send(const void *buf, size_t bytes, int peer);   // byte counts: fp64 and fp32 slabs share the transport
void comm_recv(void *buf, size_t bytes, int peer);
void comm_allgather(const double *send, double *recv, size_t count_per_rank);

// small host helper: run fn(begin,end) over [0,n) on the host threads
void parallel_for(size_t n, void (*fn)(size_t, size_t, void *), void *arg, size_t serial_below = 4096);

// A batch of instances of one fused node: the independent visits of one level that a cycle file's dataflow allows to
// run side by side (mg_cycle.cpp: every `-1` node starts from the zero field, so the sub-cycles of a W-cycle below one
// level depend on that level's F alone).  One launch carries all of them: blockIdx.y (coarse tail: blockIdx.x) picks
// the instance, whose arrays come from a table in device memory.  Type-erased: the fp64 and the fp32 kernels share it.
struct NodeBatchItem {
    const void *in, *F, *coarse;   // level-0 input (nullptr: zero / recomputed), source, coarse grid of a fused prolongation
    void *out, *Fc;                // smoothed field, next level's F of a fused restriction
};
struct NodeBatch {
    int n = 0;
    const NodeBatchItem *dev = nullptr;     // [n] in device memory
    double *const *err_outs = nullptr;      // [n] on the host: where each instance's smoothing error goes (device addresses)
};
struct TailBatchItem {
    const void *F_top;
    void *U_top;
    double *err_dev;    // the instance's error slots (node.err_slot counts from here)
    int *gs_state;
};
// One fused node launch as the cycle driver's dataflow trace records it (mg_cycle.cpp: build_schedule): while
// Context::trace is set, the fused-node entry points describe their launch here instead of enqueueing it.
struct NodeOp {
    int kind = 0;                 // 0: streaming kernel, 1: register-tile kernel, 2: coarse tail (tail = index of its TailArgs)
    int N = 0;
    double L = 1.0;
    const double *src = nullptr;  // level-0 input (nullptr: zero start, or recomputed when pre > 0)
    double *F = nullptr, *dst = nullptr;
    int take = 0;                 // sweeps
    double *err = nullptr;        // device slot of the smoothing error
    int d_sign = -1;
    const double *coarse = nullptr;
    int Nc = 0;
    double *Fc = nullptr;
    int M = 0;
    int pre = 0;
    bool no_out = false;
    int tail = -1;
    char name[48] = {0};
    double bytes = 0.0;
};

// ---------------------------------------------------------------------------
// kernel launchers (mg_kernels.hip).  All enqueue on s and return immediately.
// ---------------------------------------------------------------------------
namespace k {
// one Jacobi sweep in correction form, in -> out (in == nullptr: all zero)
void jacobi_simple(hipStream_t s, int N, double dx2, const double *in, const double *F, double *out);
// D = sign * (inv*(star - 4U) - F), rim sign*0
void residual(hipStream_t s, int N, double inv, const double *U, const double *F, double *D, int sign);
// doSmoothing's error: *out = (S+S)/N/N, S = sum over (row+col) even interior of |inv*star-F|
void smoothing_error(hipStream_t s, int N, double inv, const double *U, const double *F, double *out);
// second stage of doSmoothing's error: *out = (sum+sum)/N/N over n per-block partials
void finish_smoothing_error(hipStream_t s, const double *part, size_t n, int N, double *out);
constexpr int MAX_NORMS_PER_FLUSH = 160;   // (the descriptors travel as kernel arguments: 24 B each, 4 KB at most)
struct NormBatch {
    const double *part[MAX_NORMS_PER_FLUSH];
    double *out[MAX_NORMS_PER_FLUSH];
    int n[MAX_NORMS_PER_FLUSH];
    int N[MAX_NORMS_PER_FLUSH];
};
void finish_smoothing_errors(hipStream_t s, const NormBatch &b, int count);
// temporally blocked streaming smoother (mg_stream.hip): steps <= stream_max_steps()
int  stream_max_steps();
bool stream_supported(int N);
bool stream_fusable(int N);   // the fused prolongation / restriction stages exist for this N
// the recomputing fused `1` node (pre sweeps from zero redone in flight, then `steps` more) is instantiated for this pair
// in THIS build (it depends on the prefetch depth the library was compiled with, MG_PF)
bool stream_recompute_supported(int pre, int steps);
// coarse != nullptr: level 0 is in + doProlongation(coarse) (tables pt).  Fc != nullptr: the
// d_sign-ed residual of the result is restricted into Fc (M x M, tables rt).
// fine_w / coarse_w / fc_w: row windows of the fine arrays, the coarse input and the coarse
// output for the 1-D row-slab decomposition (nullptr = the whole grid is local).
void jacobi_stream(hipStream_t s, int N, double dx2, double inv, const double *in, const double *F,
                   double *out, int steps, double *err_out, double *D_out, int d_sign,
                   const double *coarse, int Nc, const ProlongTable *pt, double *Fc, int M,
                   const RestrictTable *rt, const RowWindow *fine_w = nullptr,
                   const RowWindow *coarse_w = nullptr, const RowWindow *fc_w = nullptr,
                   // pre > 0 (fused `1` node): `in` is not read, it is recomputed as `pre` sweeps from zero on F;
                   // no_out (fused `-1` node): the smoothed field is not stored (its `1` node will recompute it)
                   int pre = 0, bool no_out = false,
                   // batch: the same node on batch->n instances in one launch (in/F/out/coarse/Fc above then only tell the
                   // node's shape -- which of them a node of this kind has -- and err_out is ignored)
                   const NodeBatch *batch = nullptr);
// register-tile fused nodes of the small levels (mg_tile.hip / mg_tile_f32.hip): one launch = level 0 (zero | in | in +
// P(coarse)), `steps` sweeps, the error norm, optionally the d_sign-ed residual restricted into Fc; whole grid only
bool tile_wanted(int N);      // MG_TILE_MIN_N <= N <= MG_TILE_MAX_N
bool tile_wanted_slab(int N); // the same for a launch on a row window (a slab of a distributed level): up to MG_TILE_SLAB_MAX_N
int  tile_max_steps();
// fine_w / coarse_w / fc_w: row windows as in jacobi_stream (nullptr = the whole grid is local); with a fine window the error
// output is the raw sum over the counted rows
void jacobi_tile(hipStream_t s, int N, double dx2, double inv, const double *in, const double *F, double *out, int steps,
                 double *err_out, int d_sign, const double *coarse, int Nc, const ProlongTable *pt, double *Fc, int M,
                 const RestrictTable *rt, bool no_out, const RowWindow *fine_w = nullptr, const RowWindow *coarse_w = nullptr,
                 const RowWindow *fc_w = nullptr, const NodeBatch *batch = nullptr);   // batch: as in jacobi_stream
void jacobi_tile_f32(hipStream_t s, int N, float dx2, float inv, const float *in, const float *F, float *out, int steps,
                     double *err_out, int d_sign, const float *coarse, int Nc, const ProlongTable *pt, float *Fc, int M,
                     const RestrictTable *rt, bool no_out, const RowWindow *fine_w = nullptr, const RowWindow *coarse_w = nullptr,
                     const RowWindow *fc_w = nullptr);
void restrict_gather(hipStream_t s, int N, const double *Uf, int M, double *Uc, const RestrictTable &t, int sign);
// Uf_out = (Uf_in ? Uf_in : 0) + P(Uc); when Uf_in == nullptr unowned fine points are left untouched
void prolong(hipStream_t s, int N, const double *Uc, int M, const double *Uf_in, double *Uf_out, const ProlongTable &t);
void restrict_gather_f32(hipStream_t s, int N, const float *Uf, int M, float *Uc, const RestrictTable &t, int sign);
void prolong_add_f32(hipStream_t s, int N, const float *Uc, int M, const float *Uf_in, float *Uf_out, const ProlongTable &t);
void convert_to_f32(hipStream_t s, float *dst, const double *src, size_t n);
void convert_to_f64(hipStream_t s, double *dst, const float *src, size_t n);
void refine_residual(hipStream_t s, int N, double inv, const double *U, const double *F, float *src, double *err_out);
void add_widened(hipStream_t s, double *U, const float *e, size_t n);
void refine_residual_rows(hipStream_t s, int N, double inv, const double *U, const double *F, float *src, const RowWindow &w,
                          double *out_raw);
void add(hipStream_t s, size_t n, double *a, const double *b);
void negate(hipStream_t s, size_t n, double *a);
// F points at row row_lo; rows [row_lo, row_hi)
void source_device(hipStream_t s, int N, double L, double *F, double min_x, double min_y, int row_lo, int row_hi);
void analytic(hipStream_t s, int N, double L, double *U, double min_x, double min_y);
void analytic_error(hipStream_t s, int N, double L, const double *U, double min_x, double min_y, double *out);
// raw sum |analytic - U| over the owned rows of a row window (combined across slabs by the caller)
void analytic_error_rows(hipStream_t s, int N, double L, const double *U, const RowWindow &w, double min_x,
                         double min_y, double *out_raw);
void fill_uniform(hipStream_t s, double *dst, size_t n, uint64_t seed);
void checksum(hipStream_t s, const double *src, size_t n, uint64_t *out_dev /*[2]*/);
// fp32 instantiation of the streaming smoother (mg_stream_f32.hip): zero start or prolongation input,
// optional restriction output; whole grid local
void jacobi_stream_f32(hipStream_t s, int N, float dx2, float inv, const float *in, const float *F, float *out, int steps,
                       double *err_out, const float *coarse, int Nc, const ProlongTable *pt, float *Fc, int M,
                       const RestrictTable *rt, const RowWindow *fine_w = nullptr, const RowWindow *coarse_w = nullptr,
                       const RowWindow *fc_w = nullptr, double *out_wide = nullptr, float *D_out = nullptr, int d_sign = -1,
                       int pre = 0, bool no_out = false);  // pre / no_out: as in jacobi_stream
// coarse tail of a cycle in one launch (mg_tail.hip): the node slice that stays on levels N <= 64
constexpr int TAIL_MAX_LEVELS = 6;
constexpr int TAIL_MAX_NODES = 48;
constexpr int TAIL_MAX_N = 64;
struct TailNode {
    int type;      // -1, 0, 1
    int steps;     // smoothing steps (-1 / 1)
    int err_slot;  // index into err_dev, or -1
    int pad;
    double tol;    // exact-solver target (0)
};
template <typename T>
struct TailArgsT {
    int n_levels, n_nodes;
    int N[TAIL_MAX_LEVELS];
    T dx2[TAIL_MAX_LEVELS], inv[TAIL_MAX_LEVELS];
    double gs_h2[TAIL_MAX_LEVELS], gs_inv[TAIL_MAX_LEVELS];  // fp64 spacings: the exact solver is fp64 in every mode
    const int *r_lo[TAIL_MAX_LEVELS];      // restriction level l -> l+1
    const T *r_w[TAIL_MAX_LEVELS];
    const int *p_orow[TAIL_MAX_LEVELS], *p_ocol[TAIL_MAX_LEVELS];  // prolongation level l+1 -> l
    const T *p_rhi[TAIL_MAX_LEVELS], *p_rlo[TAIL_MAX_LEVELS], *p_chi[TAIL_MAX_LEVELS], *p_clo[TAIL_MAX_LEVELS];
    T c_dx[TAIL_MAX_LEVELS];
    int tab_real0, tab_int0;  // where the staged tables start in LDS (set by the launcher)
    long long *trace;         // diagnostics (MG_TAIL_TRACE): 100 MHz timestamps at start, after staging, after each node
    const T *F_top;
    T *U_top;
    double *err_dev;   // norms are fp64 whatever the field type
    int *gs_state;
    const TailBatchItem *batch;   // != nullptr: blockIdx.x picks the instance's F_top / U_top / err_dev / gs_state
    TailNode nodes[TAIL_MAX_NODES];
};
typedef TailArgsT<double> TailArgs;
typedef TailArgsT<float> TailArgsF;
void tail_launch_f32(hipStream_t s, const TailArgsF &a);
bool tail_fits(const TailArgs &a);
void tail_launch(hipStream_t s, const TailArgs &a, int n_batch = 1, const TailBatchItem *batch_dev = nullptr);
// red-black Gauss-Seidel to tolerance, fully on device; iterations -> state[1]
void gauss_seidel(hipStream_t s, int N, double h2, double inv, double *U, const double *F, double tol,
                  int *state);
void gauss_seidel_blocks_launch(hipStream_t s, int N, double h2, double inv, double *U, const double *F, double tol, int *state);
int  gs_single_workgroup_max_n();
}  // namespace k

// The pre-smoothed U of a level is dead weight between its `-1` and its `1` node: 8 B per point written, 8 B read.  When
// recompute_available(), the `-1` node (zero start) may run with smooth_restrict_no_out() and the `1` node with
// prolong_smooth_recompute(), which redoes the `pre` sweeps from zero on the same F inside its own pipeline: the same
// expressions, the same bits, two array passes less.
bool recompute_available(int Nc, int N, int pre, int step);
int  recompute_min_n();   // MG_RECOMPUTE_MIN_N (default 4096)
// launch a recorded fused node on n instances (batch == nullptr: on the arrays recorded in op itself)
void replay_node(const NodeOp &op, const NodeBatch *batch);
void smooth_restrict_no_out(int N, double L, double *U_unused, double *F, int step, double *error_dev, int M, double *F_c);
void prolong_smooth_recompute(int Nc, const double *U_c, int N, double L, double *U_out, double *F, int pre, int step, double *error_dev);
// the same pair on fp32 fields (mixed-precision mode); U_out_wide != nullptr: the result is stored in fp64 there
void smooth_restrict_f32_no_out(int N, double L, float *U_unused, float *F, int step, double *error_dev, int M, float *F_c);
void prolong_smooth_f32_recompute(int Nc, const float *U_c, int N, double L, float *U_out, double *U_out_wide, const float *F, int pre,
                                  int step, double *error_dev);
// mg_prolong_smooth_f32 whose result goes to an fp64 array (exact widening in the store) instead of U_out
void prolong_smooth_f32_wide(int Nc, const float *U_c, int N, double L, const float *U_in, double *U_out_wide, const float *F,
                             int step, double *error_dev);
// fp32 view of a coarse-tail slice: spacings and transfer weights rounded once (mg_cycle.cpp)
k::TailArgsF tail_args_f32(const k::TailArgs &a);
// collect the coarse-tail slice of a cycle file's node stream (mg_cycle.cpp)
bool scan_tail(const std::vector<double> &tokens, size_t *tok_io, const std::vector<int> &sizes, int at0, int con_step,
               int top_N, double L, k::TailArgs *out, int *node_level);

}  // namespace mg
