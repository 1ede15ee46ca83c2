/*
 * mg_oracle_driver.c -- CPU oracle of the cycle-file driver (TEST INFRASTRUCTURE,
 * see mg_oracle.h).  Restates main() of src/MG_solver_CPU.cpp:36-462: header
 * parsing, the level stack of src/linkedlist.cpp:7-124, the node semantics and the
 * printed report.  The operators are reached through an orc_ops table so the same
 * driver can run the oracle's restatement or the reference's own operators
 * (oracle/_ref) -- the latter is what bench.py times as cpu_baseline "reference".
 *
 * Deliberate deviations from the reference program (all outside its defined
 * behaviour): running off the end of the generated N_array (the reference reads out
 * of bounds) and a premature end of file are reported
 * as status != 0 instead of executing garbage; exact-solver option 0 is refused.
 */
#include "mg_oracle.h"

#include <math.h>
#include <omp.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- growable text buffer for the report -------------------------------- */
typedef struct {
    char *s;
    size_t len, cap;
} text;

static void text_printf(text *t, const char *fmt, ...)
{
    if (!t) return;
    va_list ap;
    char tmp[256];
    va_start(ap, fmt);
    int n = vsnprintf(tmp, sizeof tmp, fmt, ap);
    va_end(ap);
    if (n < 0) return;
    if (t->len + (size_t)n + 1 > t->cap) {
        t->cap = (t->cap + (size_t)n + 1) * 2;
        t->s = (char *)realloc(t->s, t->cap);
    }
    memcpy(t->s + t->len, tmp, (size_t)n + 1);
    t->len += (size_t)n;
}

/* ---- level stack: src/linkedlist.h:4-60, src/linkedlist.cpp:7-124 -------- */
typedef struct level {
    int N;
    double *U, *F, *D;
    struct level *prev;
    int step;
    double smoothingError;
} level;

typedef struct {
    level *last;
    int depth;
    int init; /* 1 until the stack has collapsed back to one level once (:63-67) */
} stack;

static void push_level(stack *st, int n)
{
    level *lv = (level *)calloc(1, sizeof(level));
    const size_t bytes = (size_t)n * n * sizeof(double);
    lv->N = n;
    lv->U = (double *)malloc(bytes);
    lv->F = (double *)malloc(bytes);
    lv->D = (double *)malloc(bytes);
    lv->prev = st->last;
    st->last = lv;
    st->depth++;
}

static void pop_level(stack *st)
{
    level *lv = st->last;
    st->last = lv->prev;
    st->depth--;
    free(lv->U);
    free(lv->F);
    free(lv->D);
    free(lv);
    if (st->depth == 1) st->init = 0;
}

static int keep_guess(const stack *st) /* U-init rule :209-214, :252-257 */
{
    return st->init == 0 && st->depth == 1;
}

/* ---- default operator table --------------------------------------------- */
static void d_getResidual(int N, double L, double *U, double *F, double *D) { orc_getResidual(N, L, U, F, D); }
static void d_doGridAddition(int N, double *a, double *b) { orc_doGridAddition(N, a, b); }
static void d_doSmoothing(int N, double L, double *U, double *F, int s, double *e) { orc_doSmoothing(N, L, U, F, s, e); }
static void d_doExactSolver(int N, double L, double *U, double *F, double t, int o) { orc_doExactSolver(N, L, U, F, t, o); }
static void d_doRestriction(int N, double *f, int M, double *c) { orc_doRestriction(N, f, M, c); }
static void d_doProlongation(int N, double *c, int M, double *f) { orc_doProlongation(N, c, M, f); }

static const orc_ops default_ops = {orc_getSource,   orc_getAnalytic,  d_getResidual,   d_doGridAddition,
                                    d_doSmoothing,   d_doExactSolver,  d_doRestriction, d_doProlongation};

static void add_record(orc_result *out, int *cap, int node, int N, int steps, double err)
{
    if (out->n_records == *cap) {
        *cap = *cap ? *cap * 2 : 32;
        out->records = (orc_node_record *)realloc(out->records, (size_t)*cap * sizeof(orc_node_record));
    }
    orc_node_record r = {node, N, steps, err};
    out->records[out->n_records++] = r;
}

/* smoothing with the error trigger, :194-230 and :376-402 (TRIGGER :99) */
static int trigger_smoothing(const orc_ops *ops, int N, double L, double *U, double *F, level *lv)
{
    const double TRIGGER = 0.01;
    double slope = TRIGGER + 1.0, before = 0;
    lv->step = 0;
    while (slope > TRIGGER) {
        ops->doSmoothing(N, L, U, F, 1, &lv->smoothingError);
        lv->step += 1;
        if (lv->step == 1) {
            before = lv->smoothingError;
            continue;
        }
        slope = fabs(lv->smoothingError - before);
        before = lv->smoothingError;
    }
    return lv->step;
}

static void report_smoothing(text *rep, int N, int steps, double err)
{
    text_printf(rep, "          ~Smoothing~\n");
    text_printf(rep, "Current Grid Size N = %d\n", N);
    text_printf(rep, "    Smoothing Steps = %d\n", steps);
    text_printf(rep, "              Error = %lf\n", err);
}

int orc_runCycleFile(const char *path, const orc_ops *ops, orc_result *out, char **report)
{
    text rep_store = {0, 0, 0};
    text *rep = report ? &rep_store : NULL;
    if (!ops) ops = &default_ops;
    memset(out, 0, sizeof *out);
    int rec_cap = 0;

    FILE *fp = fopen(path, "r");
    if (!fp) { out->status = 1; return 1; }

    double L, min_x, min_y;
    int con_step, con_N, N_max, N_min;
    if (fscanf(fp, "%lf %lf %lf %d %d %d %d", &L, &min_x, &min_y, &con_step, &con_N, &N_max, &N_min) != 7) {
        fclose(fp);
        out->status = 2;
        return 2;
    }

    /* generated level sizes, :111-146 */
    int *sizes = NULL, n_sizes = 0, at = 0;
    if (con_N == 1) {
        for (int n = N_max; n >= N_min; n /= 2) n_sizes++;
        sizes = (int *)malloc((size_t)(n_sizes + 1) * sizeof(int));
        int n = N_max;
        for (int i = 0; i < n_sizes; ++i, n /= 2) sizes[i] = n;
    } else if (con_N == 2) {
        n_sizes = N_max - N_min + 1;
        sizes = (int *)malloc((size_t)(n_sizes + 1) * sizeof(int));
        for (int i = 0; i < n_sizes; ++i) sizes[i] = N_max - i;
    }

    stack st = {NULL, 0, 1};
    push_level(&st, N_max); /* :149 */
    ops->getSource(st.last->N, L, st.last->F, min_x, min_y); /* :153 */

    int status = 0;
    const double t0 = omp_get_wtime(); /* :156 */
    for (;;) {
        int node;
        if (fscanf(fp, "%d", &node) != 1) break; /* end of file without a 2 */
        if (node == 2) break;                     /* :162-164 */

        if (node == -1) { /* smooth, residual, restrict: :169-300 */
            int step = 0, next_N = 0;
            if (con_step == 0) { if (fscanf(fp, "%d", &step) != 1) { status = 3; break; } }
            else step = con_step;
            if (con_N == 0) { if (fscanf(fp, "%d", &next_N) != 1) { status = 3; break; } }
            else {
                if (at + 1 >= n_sizes) { status = 4; break; } /* reference: out-of-bounds read */
                next_N = sizes[++at];
            }
            if (step == 0) continue; /* :241-243, :296-299 */

            level *lv = st.last;
            const int N = lv->N;
            if (!keep_guess(&st)) memset(lv->U, 0, (size_t)N * N * sizeof(double));
            int done;
            if (step == -1) done = trigger_smoothing(ops, N, L, lv->U, lv->F, lv);
            else { ops->doSmoothing(N, L, lv->U, lv->F, step, &lv->smoothingError); done = step; }
            report_smoothing(rep, N, done, lv->smoothingError);
            add_record(out, &rec_cap, -1, N, done, lv->smoothingError);

            ops->getResidual(N, L, lv->U, lv->F, lv->D); /* :239, :268 */
            {
                double *D = lv->D;
                const size_t n = (size_t)N * N;
#pragma omp parallel for
                for (size_t i = 0; i < n; ++i) D[i] = -D[i]; /* :277-280 */
            }
            push_level(&st, next_N);                               /* :283 */
            ops->doRestriction(N, lv->D, next_N, st.last->F);      /* :287 */
            text_printf(rep, "             *\n             |\n Restriction |\n             |\n             *\n");
        } else if (node == 0) { /* :305-324 */
            double tol;
            int option;
            if (fscanf(fp, "%lf %d", &tol, &option) != 2) { status = 3; break; }
            if (option != 1) { status = 5; break; }
            level *lv = st.last;
            ops->doExactSolver(lv->N, L, lv->U, lv->F, tol, option);
            text_printf(rep, "          ~Exact Solver~\n");
            text_printf(rep, "Current Grid Size N = %d\n", lv->N);
            text_printf(rep, "   Use Exact Solver = GaussSeidel Even / Odd\n");
            text_printf(rep, "       Target Error = %.3e\n", tol);
            add_record(out, &rec_cap, 0, lv->N, 0, 0.0);
        } else if (node == 1) { /* prolong, add, smooth: :329-424 */
            int step;
            if (con_step == 0) { if (fscanf(fp, "%d", &step) != 1) { status = 3; break; } }
            else step = con_step;
            if (con_N != 0) at--;
            if (st.depth < 2) { status = 6; break; }

            level *coarse = st.last;
            const int fine_N = coarse->prev->N;
            double *tmp = (double *)malloc((size_t)fine_N * fine_N * sizeof(double)); /* :353 */
            ops->doProlongation(coarse->N, coarse->U, fine_N, tmp);                    /* :354 */
            text_printf(rep, "             *\n             |\nProlongation |\n             |\n             *\n");
            pop_level(&st);                                                            /* :363 */
            level *lv = st.last;
            ops->doGridAddition(lv->N, lv->U, tmp);                                    /* :368 */
            free(tmp);

            if (step == 0) continue;
            int done;
            if (step == -1) done = trigger_smoothing(ops, lv->N, L, lv->U, lv->F, lv);
            else { ops->doSmoothing(lv->N, L, lv->U, lv->F, step, &lv->smoothingError); done = step; }
            report_smoothing(rep, lv->N, done, lv->smoothingError);
            add_record(out, &rec_cap, 1, lv->N, done, lv->smoothingError);
        }
        /* any other token is ignored, as in the reference */
    }
    const double t1 = omp_get_wtime(); /* :429 */
    fclose(fp);

    /* final error against the analytic solution, :434-445 (serial, in index order) */
    level *lv = st.last;
    const int N = lv->N;
    const size_t n = (size_t)N * N;
    double *ana = (double *)malloc(n * sizeof(double));
    ops->getAnalytic(N, L, ana, min_x, min_y);
    double acc = 0.0;
    for (size_t i = 0; i < n; ++i) acc = acc + fabs(ana[i] - lv->U[i]);
    free(ana);

    out->N = N;
    out->U = (double *)malloc(n * sizeof(double));
    memcpy(out->U, lv->U, n * sizeof(double));
    out->mg_error = acc / (double)(N * N);
    out->time_ms = 1000.0 * (t1 - t0);
    out->status = status;
    text_printf(rep, "\n\n===== Final Result =====\n    Error = %lf\n", out->mg_error);

    while (st.depth > 0) pop_level(&st);
    free(sizes);
    if (report) *report = rep_store.s;
    return status;
}

/* src/MG_solver_CPU.cpp:735-754: top row first, "%lf" with ',' separators */
int orc_print2File(int N, const double *U, const char *file_name)
{
    FILE *o = fopen(file_name, "w");
    if (!o) return 1;
    for (int r = N - 1; r >= 0; --r)
        for (int c = 0; c < N; ++c) fprintf(o, c == N - 1 ? "%lf\n" : "%lf,", U[c + (size_t)N * r]);
    fclose(o);
    return 0;
}
