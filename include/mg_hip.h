/*
 * mg_hip.h -- C ABI of the MI355X-native multigrid Poisson engine (libmgpoisson.so).
 *
 * Drop-in boundary for the V/W-cycle hot path of cindytsai/multigrid_poisson_solver.
 * The reference has no FFI layer: its "interface" is the set of free functions its
 * driver calls (src/MG_solver_CPU.cpp:16-30, GPU twins src/MG_solver_GPU.cu:27-41).
 * Every operator below keeps the reference's argument list and meaning; the one
 * change is that U/F/D are DEVICE-resident fp64 arrays (row-major N x N,
 * index = col + N*row, boundary included -- src/MG_solver_CPU.cpp:484) obtained from
 * mg_alloc(), because re-uploading per call (src/MG_solver_GPU.cu:1159-1282) can never
 * approach the HBM roofline.  include/mg_dropin.hpp maps the reference's C++ names
 * onto these symbols so the reference's own main() compiles against this library
 * (see INTEGRATION.md).
 *
 * Conventions kept from the reference: caller owns all arrays,
 * callee may use internal scratch, results are visible when the call returns for every
 * entry point that hands a value back to the host, single caller thread, int sizes.
 * Errors: the reference printf+exit(1)s; here a failed call prints to stderr, records
 * a code readable through mg_last_error() and (default) aborts the process with
 * exit(1) -- mg_set_abort_on_error(0) turns the abort off for embedding in tests.
 *
 * All calls are enqueued on ONE HIP stream (mg_set_stream / mg_get_stream).
 */
#ifndef MG_HIP_H
#define MG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------- */
/* lifecycle                                                                  */
/* ------------------------------------------------------------------------- */
/* replaces cudaSetDevice(0), src/MG_solver_GPU.cu:58-62.  Returns 0 on success. */
int         mg_init(int device);
void        mg_finalize(void);
/* use the caller's hipStream_t (NULL = the engine's own stream) */
void        mg_set_stream(void *hip_stream);
void       *mg_get_stream(void);
void        mg_sync(void);
int         mg_last_error(void);
const char *mg_last_error_string(void);
void        mg_clear_error(void);
void        mg_set_abort_on_error(int on);
/* "stream" (default: temporally blocked wave-streaming kernels) or "simple"
 * (one launch per sweep, one thread per point); also read from env MG_SMOOTHER. */
int         mg_set_smoother(const char *name);
const char *mg_version(void);

/* ------------------------------------------------------------------------- */
/* memory: replaces malloc/free of ListNode (src/linkedlist.cpp:9-11,21-23,48-50)
 * and of tempU (src/MG_solver_CPU.cpp:353,371).  Served from a caching pool so the
 * timed window contains no hipMalloc.                                         */
/* ------------------------------------------------------------------------- */
double *mg_alloc(size_t n_doubles);
void    mg_free(double *dev);
void    mg_pool_trim(void);
size_t  mg_pool_bytes(void);
void    mg_upload(double *dev, const double *host, size_t n_doubles);   /* synchronous */
void    mg_download(double *host, const double *dev, size_t n_doubles); /* synchronous */
void    mg_copy(double *dst_dev, const double *src_dev, size_t n_doubles);
/* memset(U, 0, N*N*8): src/MG_solver_CPU.cpp:213,256 */
void    mg_fill_zero(double *dev, size_t n_doubles);
/* D[i] = -D[i] for all N*N entries: the driver's own loop, src/MG_solver_CPU.cpp:277-280 */
void    mg_negate(int N, double *D);

/* ------------------------------------------------------------------------- */
/* problem definition: src/MG_solver_CPU.cpp:468-548                          */
/* ------------------------------------------------------------------------- */
void mg_getSource(int N, double L, double *F, double min_x, double min_y);
/* where mg_getSource (and the drivers' own getSource at load, :153) evaluates F:
 * "auto" (default): on the device (k_source: no host pass, no PCIe) when that reproduces THIS host's libm bit for
 *   bit -- the reference calls libm's exp() (:488), and the device evaluates glibc's algorithm for it (table +
 *   polynomial, FMA form); the engine compares the two on ~130k points once per process (mg_source_is_bit_identical)
 *   and falls back to "host" when they differ (another libm, a CPU without FMA), and per call for grids whose
 *   arguments x - y leave the verified main path of the algorithm (|min_x - min_y| + |L| >= 511: glibc's overflow
 *   and subnormal special cases are not reproduced on the device).  mg_getAnalytic / mg_analyticError always use the
 *   device form: outside that range their exp() is the device's own (<= 1 ulp from libm's);
 * "host": libm on the host cores, chunked through 2 x 128 MiB of pinned staging;
 * "device": the device form unconditionally.  Also env MG_SOURCE. */
int  mg_set_source(const char *mode);
/* "host" or "device": what mg_getSource will use (runs the check in auto mode) */
const char *mg_source_mode(void);
int  mg_source_is_bit_identical(void);
void mg_getAnalytic(int N, double L, double *U, double min_x, double min_y);
/* sum|analytic - U| / (N*N): src/MG_solver_CPU.cpp:434-445; result to *error_host */
void mg_analyticError(int N, double L, const double *U, double min_x, double min_y, double *error_host);

/* ------------------------------------------------------------------------- */
/* the six operators: src/MG_solver_CPU.cpp:23-28 (same order, same arguments)  */
/* ------------------------------------------------------------------------- */
/* :554-564 */
void mg_getResidual(int N, double L, double *U, double *F, double *D);
/* :566-571 */
void mg_doGridAddition(int N, double *U1, double *U2);
/* :573-625.  error is a HOST pointer (as in the reference, where it points into the
 * ListNode); the call returns after the value is written.  NULL = not wanted. */
void mg_doSmoothing(int N, double L, double *U, double *F, int step, double *error);
/* :627-638, option 1 = red-black Gauss-Seidel :952-1066.  option 0 (dense inverse) is
 * refused with the message of src/MG_solver_GPU.cu:1286-1289. */
void mg_doExactSolver(int N, double L, double *U, double *F, double target_error, int option);
/* :640-680 */
void mg_doRestriction(int N, double *U_f, int M, double *U_c);
/* :682-724 */
void mg_doProlongation(int N, double *U_c, int M, double *U_f);

/* ------------------------------------------------------------------------- */
/* asynchronous / fused forms used by the engine's own driver                 */
/* ------------------------------------------------------------------------- */
/* step Jacobi sweeps from U_in into U_out (U_out != U_in; U_in is clobbered as
 * scratch when more than one launch is needed).  U_in == NULL means "U is all zero"
 * (the driver's memset of src/MG_solver_CPU.cpp:256 folded into the first sweep).
 * error_dev (device pointer or NULL) receives doSmoothing's error; D_out (device
 * pointer or NULL) receives getResidual(U_out) times d_sign (+1 or -1: the driver's
 * sign flip :277-280 folded in). */
void mg_smooth_pp(int N, double L, const double *U_in, double *U_out, double *F, int step,
                  double *error_dev, double *D_out, int d_sign);
/* one "-1" node of the driver (:259-287): U_out = smooth^step(U_in or 0), then
 * F_c = doRestriction(N, -getResidual(U_out), M) -- one pass over HBM when fusable */
void mg_smooth_restrict(int N, double L, const double *U_in, double *U_out, double *F, int step,
                        double *error_dev, int M, double *F_c);
/* one "1" node of the driver (:353-416): U_out = smooth^step(U_in + doProlongation(Nc, U_c, N)) */
void mg_prolong_smooth(int Nc, const double *U_c, int N, double L, const double *U_in, double *U_out,
                       double *F, int step, double *error_dev);
/* mixed-precision mode (README.md:269-270 of the reference: "single
 * precision" GPU kernels): the two fused nodes with every array and every arithmetic operation
 * in fp32 (norms in fp64).  Zero start only for the "-1" node; 1..4 steps.  Even N with a nested
 * coarse size runs as ONE launch, anything else operator by operator (same results). */
void   mg_smooth_restrict_f32(int N, double L, const float *U_in, float *U_out, float *F, int step,
                              double *error_dev, int M, float *F_c);
void   mg_prolong_smooth_f32(int Nc, const float *U_c, int N, double L, const float *U_in, float *U_out,
                             float *F, int step, double *error_dev);
float *mg_alloc_f32(size_t n_floats);
void   mg_free_f32(float *dev);
void   mg_to_f32(float *dst_dev, const double *src_dev, size_t n);   /* round to nearest */
void   mg_to_f64(double *dst_dev, const float *src_dev, size_t n);   /* exact */
void   mg_upload_f32(float *dev, const float *host, size_t n);       /* synchronous */
void   mg_download_f32(float *host, const float *dev, size_t n);     /* synchronous */
/* U_f_out = U_f_in + doProlongation(N, U_c, M) in one pass (:354 + :368) */
void mg_prolongAdd(int N, const double *U_c, int M, const double *U_f_in, double *U_f_out);
/* doRestriction(N, sign*U_f, M, U_c) */
void mg_restrict_signed(int N, const double *U_f, int M, double *U_c, int sign);
/* number of Gauss-Seidel iterations the last mg_doExactSolver call ran (synchronises) */
int  mg_lastExactSolverIterations(void);

/* ------------------------------------------------------------------------- */
/* host-side index tables (exact reference expressions, fp64 on the host)      */
/* ------------------------------------------------------------------------- */
/* doRestriction :661-666: lo[i] = (int)floor(i*h_c/h_f), w[i] = fmod(i*h_c,h_f)/h_f */
void mg_restriction_table(int N, int M, int *lo, double *w);
/* doProlongation :697-718: owner of fine index k along rows (axis 0) or columns
 * (axis 1) and the two 1-D weights (c_hi - f, f - c_lo) the kernel multiplies with */
void mg_prolongation_table(int N, int M, int axis, int *owner, double *w_hi, double *w_lo);

/* ------------------------------------------------------------------------- */
/* synthetic inputs and checksums (benchmark + full-size parity tests)         */
/* ------------------------------------------------------------------------- */
/* dst[i] = uniform[0,1) from a counter-based 64-bit hash of (i + seed); the same
 * recipe as tests/_synth.py:hash_uniform */
void mg_fill_uniform(double *dst, size_t n_doubles, uint64_t seed);
/* out[0] = sum of bit patterns, out[1] = sum of bit patterns * (2*i+1), mod 2^64,
 * with -0.0 canonicalised to +0.0 (tests/_synth.py:checksum); synchronous */
void mg_checksum(const double *src, size_t n_doubles, uint64_t out[2]);

/* ------------------------------------------------------------------------- */
/* live kernel timing (bench.py's roofline object): hipEvent pairs recorded on   */
/* the engine's stream around every operator launch on grids with N >= min_N     */
/* ------------------------------------------------------------------------- */
typedef struct mg_profile_entry {
    char   name[64];        /* kernel family, e.g. "jacobi_stream<3>" */
    int    N;               /* grid size */
    int    launches;
    double total_ms;        /* sum of the launches' durations */
    double algo_bytes;      /* ALGORITHMIC bytes of ONE launch */
} mg_profile_entry;
void mg_profile_begin(int min_N);
/* time the launches of every `every`-th cycle window (mg_cycle_enqueue) only -- the 1st, the (every+1)-th, ...; 1 = all */
void mg_profile_sample(int every);
/* synchronises, fills out[0..cap) with one entry per (name, N), returns the count */
int  mg_profile_end(mg_profile_entry *out, int cap);

/* ------------------------------------------------------------------------- */
/* cycle-file driver: main() of src/MG_solver_CPU.cpp:36-462                   */
/* ------------------------------------------------------------------------- */
typedef struct mg_node_record {
    int    node;   /* -1, 0, 1 */
    int    N;
    int    steps;
    double error;
} mg_node_record;

typedef struct mg_cycle_result {
    int     status;        /* 0 ok */
    int     N;             /* finest grid */
    double *U_dev;         /* final solution (device, owned by the plan) */
    double  mg_error;      /* :434-445 */
    double  time_ms;       /* host wall clock of the reference's window :156..:429, synchronised */
    double  device_ms;     /* hipEvent time of the same window */
    int     n_records;
    const mg_node_record *records; /* owned by the plan, valid until the next execute */
    const char *report;    /* the reference's printed report (owned by the plan) */
    int     graph_replayed;     /* 1: the last window was a hipGraph replay (MG_CYCLE_GRAPH took effect), 0: launched eagerly */
    int     schedule_launches;  /* > 0: the window ran as a batched breadth-first schedule of this many launches (the
                                 * independent visits of a level merged into one launch each); 0: node by node */
} mg_cycle_result;

typedef struct mg_cycle_plan mg_cycle_plan;

#define MG_CYCLE_FUSED   1 /* use the fused operators (default driver mode) */
#define MG_CYCLE_GRAPH   2 /* capture the node program into a hipGraph and replay it */
#define MG_CYCLE_REPORT  4 /* build the printed report text */
#define MG_CYCLE_ERROR   8 /* evaluate mg_error (:434-445) after the window, as the program does */
#define MG_CYCLE_MIXED  16 /* the whole cycle in fp32 (F rounded once at load, U widened at the end);
                              fixed-step halving files whose coarse part fits the tail kernel */

/* parse a cycle structure file (README.md:43-128); allocates the finest level and
 * evaluates getSource on it (:149-153, outside the timed window) */
mg_cycle_plan *mg_cycle_load(const char *path, int flags);
/* one run of the reference's timed window from the state right after getSource */
int            mg_cycle_execute(mg_cycle_plan *plan, mg_cycle_result *out);
/* the same window without host synchronisation: enqueue any number back to back, then collect
 * (waits, reports the last one).  mg_cycle_execute == sync + enqueue + sync + collect. */
int            mg_cycle_enqueue(mg_cycle_plan *plan);
int            mg_cycle_collect(mg_cycle_plan *plan, mg_cycle_result *out);
/* MG_CYCLE_MIXED plans: `cycles` fp32 runs of the file per window, joined by the fp64 residual of
 * the fp64 iterate and an fp64 correction (iterative refinement; BASELINE.json configs[4]).  The
 * errors are doSmoothing's metric (src/MG_solver_CPU.cpp:607-622) of the iterate after 1 .. cycles-1
 * corrections. */
int            mg_cycle_set_refinement(mg_cycle_plan *plan, int cycles);
int            mg_cycle_refinement_errors(mg_cycle_plan *plan, double *out, int cap);
void           mg_cycle_destroy(mg_cycle_plan *plan);
/* the whole reference program: load, execute, print report, write Sol_HIP_<file> CSV */
int            mg_cycle_main(int argc, char **argv);
/* CSV writer of src/MG_solver_CPU.cpp:735-754 on a device array */
int            mg_print2File(int N, const double *U_dev, const char *file_name);

/* ------------------------------------------------------------------------- */
/* 1-D row-slab decomposition over the GPUs of one node (no reference            */
/* counterpart: the reference is single-device, src/MG_solver_GPU.cu:58)         */
/* ------------------------------------------------------------------------- */
/* RCCL communicator, one process per GPU.  Rank 0 creates the id, the caller ships its
 * mg_comm_unique_id_bytes() bytes to every rank (e.g. torch.distributed broadcast). */
int  mg_comm_unique_id_bytes(void);
int  mg_comm_get_unique_id(void *out);
int  mg_comm_init(int rank, int nranks, const void *unique_id);
/* Host-staged transport instead of RCCL: the slab driver's ghost-row groups and its all-gather are
 * handed to two callbacks of the embedding program (MPI, gloo, ...) with HOST buffers.  `exchange`
 * receives the n_ops point-to-point operations of one group (is_send[i], peer[i], buf[i],
 * bytes[i]) and returns 0 when ALL of them have completed; operations between two ranks
 * are posted in the same order on both sides.  `allgather` fills recv[r*count .. ] with rank r's
 * `send`.  Purpose: running the real rank-mode driver with several processes on one GPU (RCCL
 * allows one rank per device), and clusters without xGMI.  Arithmetic stays on the device. */
typedef struct mg_host_transport {
    void *user;
    int (*exchange)(void *user, int n_ops, const int *is_send, const int *peer, void *const *buf, const size_t *bytes);
    int (*allgather)(void *user, const double *send, double *recv, size_t count_per_rank);
} mg_host_transport;
int  mg_comm_init_host(int rank, int nranks, const mg_host_transport *transport);
/* exercises the RCCL calls of the slab driver (grouped send/recv on a second stream ordered by events, all-gather)
 * against this rank itself and checks the bytes; 0 = ok.  Runs with a 1-rank communicator on a one-GPU box. */
int  mg_comm_selftest(size_t n_doubles);
void mg_comm_finalize(void);
int  mg_comm_rank(void);
int  mg_comm_size(void);
/* the shared object the RCCL entry points in use were resolved from (absolute path as the loader mapped it), "host
 * transport ..." after mg_comm_init_host, "" before any communicator exists: lets a benchmark record prove which wire
 * and how many ranks (mg_comm_size) it ran on */
const char *mg_comm_library(void);

/* host-only: row ranges per level and rank of the hierarchy N_max, N_max/2, ... >= N_min.
 * out[(level*nranks + rank)*2 + {0,1}] = [lo, hi); collapsed_out[level] = 1 where the level
 * is collapsed (replicated whole on every rank; reported as [0, N) for rank 0 and [0, 0) for the
 * others).  Returns the number of levels. */
int  mg_slab_partition(int N_max, int N_min, int nranks, int collapse_N, int *out, int *collapsed_out);
/* a level stays distributed while every slab keeps at least 2 * mg_slab_ghost_rows() rows */
int  mg_slab_ghost_rows(void);
/* host-only: the communication-avoiding schedule mg_slab_load derives for a hierarchy (`steps` sweeps per node).
 * Per distributed level and rank: own = owned rows; dext = rows its `-1` launch updates (owned rows + the rows whose
 * restricted residual is the next level's F halo, recomputed instead of exchanged); ext = rows its `1` launch updates
 * (owned rows + what the next finer level's prolongation reads); fwr = rows of the level's F the finer level's launch
 * writes.  Per level: halo rows per side, needF (F rows read beyond the owned ones), xF / xU (rows of the level's F /
 * U halo that DO travel: xF on the critical path right after the finer level's `-1` launch, xU on a second stream,
 * needed only when the cycle comes back up through the level).
 *   ca_mode 0: exchange every halo (one group per level); 1 (default): recompute F halos while the extra rows stay
 *   below ca_pct per cent of a slab (default 10); 2: recompute U halos too.  Negative = env MG_SLAB_CA / MG_SLAB_CA_PCT.
 * level_out[l*6 ..] = {N, collapsed, halo, needF, xF, xU}; rank_out[(l*nranks + r)*8 ..] = {own, dext, ext, fwr} as
 * [lo, hi) pairs.  Returns the number of levels, -1 when a halo would not fit the neighbouring slab. */
int  mg_slab_schedule(int N_max, int N_min, int nranks, int collapse_N, int steps, int ca_mode, int ca_pct,
                      int *level_out, int *rank_out);
/* host-only: pre_out[l] > 0 = on level l (only distributed levels matter) the `-1` launch does not store the smoothed
 * U and the `1` launch recomputes it, pre_out[l] sweeps from zero on F, instead of reading it (fp64 plans, levels of
 * at least MG_RECOMPUTE_MIN_N = 4096 points per side; MG_SLAB_RECOMPUTE=0 switches it off): such a level has no U
 * halo.  Returns the number of levels. */
int  mg_slab_recompute_levels(int N_max, int N_min, int steps, int *pre_out);
/* the same for a plan of `nranks` slabs: what decides is the size of a launch -- N * (N / nranks) >= MG_RECOMPUTE_MIN_N^2 / 2
 * points -- so the smaller distributed levels of a many-rank plan store and re-read their U (and exchange its halo) */
int  mg_slab_recompute_levels_ranks(int N_max, int N_min, int steps, int nranks, int *pre_out);
/* host-only: 1 when this build holds the recomputing fused `1` node for `pre` pre-smoothing + `post` post-smoothing
 * sweeps (1+1, 2+2, 3+3 in the default build; none in a build with another prefetch depth): the node pair that
 * neither stores nor re-reads a level's pre-smoothed U (src/MG_solver_CPU.cpp:259 ... :416 of one level).  Other
 * sweep counts -- and cycle files with per-node step counts, con_step = 0 -- run the store/re-read form. */
int  mg_recompute_pair_available(int pre, int post);

typedef struct mg_slab_plan mg_slab_plan;
/* rank >= 0: this process is that rank (needs mg_comm_init when nranks > 1).  rank == -1:
 * all nranks slabs live in this process and exchange by device copies ("virtual ranks":
 * how the decomposition is verified bit for bit on one GPU).  Levels with N <= collapse_N
 * are replicated: every rank runs them on the whole grid.  Supported grammar: con_N = 1, fixed con_step in 1..4, option 1. */
mg_slab_plan *mg_slab_load(const char *path, int nranks, int rank, int collapse_N);
/* the same with flags: MG_CYCLE_MIXED = the whole cycle on fp32 slabs (half the HBM and xGMI bytes;
 * source rounded once, exact solver in fp64, mg_slab_gather_U widens), BASELINE.json configs[4] */
mg_slab_plan *mg_slab_load_flags(const char *path, int nranks, int rank, int collapse_N, int flags);
/* MG_CYCLE_MIXED slab plans: like mg_cycle_set_refinement / mg_cycle_refinement_errors.  Each extra cycle
 * costs one ghost exchange of the fp64 iterate and one of the new fp32 source on the finest level. */
int  mg_slab_set_refinement(mg_slab_plan *plan, int cycles);
int  mg_slab_refinement_errors(mg_slab_plan *plan, double *out, int cap);
/* one run of the reference's timed window; U_dev is NULL (use mg_slab_gather_U) */
int  mg_slab_execute(mg_slab_plan *plan, mg_cycle_result *out);
/* like mg_cycle_enqueue / mg_cycle_collect */
int  mg_slab_enqueue(mg_slab_plan *plan);
int  mg_slab_collect(mg_slab_plan *plan, mg_cycle_result *out);
/* owned rows of this process's slabs of the finest U into a full N x N host array */
int  mg_slab_gather_U(mg_slab_plan *plan, double *host_full);
/* mg_error (src/MG_solver_CPU.cpp:434-445) is evaluated after each window unless turned off */
void mg_slab_want_error(mg_slab_plan *plan, int on);
void mg_slab_destroy(mg_slab_plan *plan);

#ifdef __cplusplus
}
#endif
#endif /* MG_HIP_H */
