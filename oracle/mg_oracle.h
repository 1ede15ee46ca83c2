/*
 * mg_oracle.h -- CPU oracle for the multigrid V/W-cycle hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library, and only as the checker / the reported CPU
 * baseline.  The product path (multigrid_poisson_solver_amd/) never links,
 * imports or calls it.
 *
 * What it is: a from-scratch fp64 restatement (plain C + OpenMP) of the
 * operators of the reference CPU program, keeping the floating-point
 * evaluation order of every expression so the output arrays are
 * bit-identical.  Each function cites the reference lines it follows
 * (paths relative to /root/reference).
 *
 * Parity pin: PINNED.  tests/test_oracle_pin.py compares every function
 * below bit-for-bit with the reference's own code compiled from
 * /root/reference/src (oracle/_ref/libmgref.so, see oracle/Makefile) when
 * that build is present, and always against the committed golden vectors
 * under tests/golden/ that were generated from that build by
 * tests/golden/make_golden.py.
 */
#ifndef MG_ORACLE_H
#define MG_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* problem definition: src/MG_solver_CPU.cpp:468-548 */
void orc_getSource(int N, double L, double *F, double min_x, double min_y);
void orc_getAnalytic(int N, double L, double *U, double min_x, double min_y);

/* operators: src/MG_solver_CPU.cpp:554-724, 952-1066 */
void orc_getResidual(int N, double L, const double *U, const double *F, double *D);
void orc_doGridAddition(int N, double *U1, const double *U2);
void orc_doSmoothing(int N, double L, double *U, const double *F, int step, double *error);
void orc_doExactSolver(int N, double L, double *U, const double *F, double target_error, int option);
void orc_doRestriction(int N, const double *U_f, int M, double *U_c);
void orc_doProlongation(int N, const double *U_c, int M, double *U_f);
/* number of Gauss-Seidel iterations the last orc_doExactSolver call ran */
int  orc_lastExactSolverIterations(void);

/* 1-D gather tables of doRestriction (src/MG_solver_CPU.cpp:661-666):
 * for coarse index i in [0,M): lo[i] = (int)floor(i*h_c/h_f), w[i] = fmod(i*h_c,h_f)/h_f */
void orc_restrictionTable(int N, int M, int *lo, double *w);
/* 1-D ownership table of doProlongation (src/MG_solver_CPU.cpp:697-698,701-718):
 * owner[k] = coarse cell i whose [ceil(i*ratio), ceil((i+1)*ratio)) range holds fine
 * index k, with the forced last row/column (k == M-1) owned by the cell that holds
 * M-2; -1 when no cell writes k. */
void orc_prolongationOwner(int N, int M, int *owner);

/* set OpenMP threads used by the oracle */
void orc_setThreads(int n);
int  orc_maxThreads(void);

/* ------------------------------------------------------------------ */
/* cycle-file driver: restates main() of src/MG_solver_CPU.cpp:36-462  */
/* ------------------------------------------------------------------ */
typedef struct orc_ops {
    void (*getSource)(int, double, double *, double, double);
    void (*getAnalytic)(int, double, double *, double, double);
    void (*getResidual)(int, double, double *, double *, double *);
    void (*doGridAddition)(int, double *, double *);
    void (*doSmoothing)(int, double, double *, double *, int, double *);
    void (*doExactSolver)(int, double, double *, double *, double, int);
    void (*doRestriction)(int, double *, int, double *);
    void (*doProlongation)(int, double *, int, double *);
} orc_ops;

/* one record per executed node, in execution order */
typedef struct orc_node_record {
    int    node;      /* -1, 0, 1 */
    int    N;         /* grid size the node worked on (the finer one for 1) */
    int    steps;     /* smoothing steps executed (0 for node 0) */
    double error;     /* smoothing error reported (0 for node 0) */
} orc_node_record;

typedef struct orc_result {
    int     N;              /* size of the final grid */
    double *U;              /* final solution, N*N, malloc'ed (caller frees with orc_free) */
    double  mg_error;       /* sum|analytic-U|/N^2   :434-445 */
    double  time_ms;        /* the reference's timed window :156..:429 */
    int     n_records;
    orc_node_record *records; /* malloc'ed */
    int     status;         /* 0 ok, nonzero = malformed cycle file */
} orc_result;

/* ops == NULL -> the oracle's own operators.  report may be NULL; when given it
 * receives exactly the text the reference program prints between "OpenMP threads"
 * and the "Output file name" line (without those two, and without "Time Used"). */
int  orc_runCycleFile(const char *path, const orc_ops *ops, orc_result *out, char **report);
void orc_free(void *p);
/* CSV writer, src/MG_solver_CPU.cpp:735-754 */
int  orc_print2File(int N, const double *U, const char *file_name);

#ifdef __cplusplus
}
#endif
#endif
