/*
 * mg_oracle.c -- CPU oracle (TEST INFRASTRUCTURE, see mg_oracle.h).
 *
 * fp64 restatement of the reference operators.  Array layout everywhere:
 * row-major N x N, index = col + N*row, boundary included
 * (src/MG_solver_CPU.cpp:484).  Floating-point expressions keep the
 * reference's association order; build with -ffp-contract=off and without
 * -ffast-math (the reference build has no FMA: src/Makefile:8).
 *
 * Parity pin: PINNED against oracle/_ref (the reference sources compiled where
 * they lie) and tests/golden/ -- see mg_oracle.h.
 */
#include "mg_oracle.h"

#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

void orc_setThreads(int n) { if (n > 0) omp_set_num_threads(n); }
int  orc_maxThreads(void) { return omp_get_max_threads(); }
void orc_free(void *p) { free(p); }

/* pow(dx, 2) of the reference (src/MG_solver_CPU.cpp:560,590,611,1020).  An optimising
 * build of the reference (gcc -O1 and up, clang) folds pow(x, 2) to x*x; the shipped
 * Makefile has no -O and calls glibc's pow(), which differs from x*x by one ulp for 36
 * grid sizes below 40000 (the smallest are 2948, 3504, 5378) and for none of the sizes
 * any shipped cycle file or BASELINE.json config generates (8 ... 32768 by halving, 16,
 * 128, 256) -- tests/test_oracle_pin.py::test_pow_is_square pins exactly that.  The
 * oracle, oracle/_ref (built -O2) and the engine all use x*x. */
static inline double square(double x) { return x * x; }

static inline int on_rim(int r, int c, int N)
{
    return r == 0 || c == 0 || r == N - 1 || c == N - 1;
}

/* ---------------------------------------------------------------------- */
/* problem definition                                                      */
/* ---------------------------------------------------------------------- */

/* src/MG_solver_CPU.cpp:468-493 (+ getBoundary :497-523: whole array zeroed,
 * rim stays 0).  f = 2x(y-1)(y-2x+xy+2)e^{x-y}, evaluated left to right (:488). */
void orc_getSource(int N, double L, double *F, double min_x, double min_y)
{
    const double h = L / (double)(N - 1);
#pragma omp parallel for
    for (int r = 0; r < N; ++r) {
        for (int c = 0; c < N; ++c) {
            double v = 0.0;
            if (!on_rim(r, c, N)) {
                const double x = (double)c * h + min_x;
                const double y = (double)r * h + min_y;
                v = 2.0 * x * (y - 1) * (y - 2.0 * x + x * y + 2.0) * exp(x - y);
            }
            F[(size_t)r * N + c] = v;
        }
    }
}

/* src/MG_solver_CPU.cpp:525-548: u = e^{x-y} x(1-x) y(1-y), rim 0 (:544). */
void orc_getAnalytic(int N, double L, double *U, double min_x, double min_y)
{
    const double h = L / (double)(N - 1);
#pragma omp parallel for
    for (int r = 0; r < N; ++r) {
        for (int c = 0; c < N; ++c) {
            double v = 0.0;
            if (!on_rim(r, c, N)) {
                const double x = (double)c * h + min_x;
                const double y = (double)r * h + min_y;
                v = exp(x - y) * x * (1.0 - x) * y * (1.0 - y);
            }
            U[(size_t)r * N + c] = v;
        }
    }
}

/* ---------------------------------------------------------------------- */
/* operators                                                               */
/* ---------------------------------------------------------------------- */

/* 5-point sum in the reference's order: row+1, row-1, col+1, col-1, then -4*centre
 * (src/MG_solver_CPU.cpp:560, :590, :611). */
static inline double star_minus4(const double *A, size_t p, int N)
{
    return A[p + N] + A[p - N] + A[p + 1] + A[p - 1] - 4 * A[p];
}

/* src/MG_solver_CPU.cpp:554-564 */
void orc_getResidual(int N, double L, const double *U, const double *F, double *D)
{
    const double dx = L / (double)(N - 1);
    const double inv = 1.0 / square(dx);
#pragma omp parallel for
    for (int r = 0; r < N; ++r) {
        for (int c = 0; c < N; ++c) {
            const size_t p = (size_t)r * N + c;
            D[p] = on_rim(r, c, N) ? 0.0 : inv * star_minus4(U, p, N) - F[p];
        }
    }
}

/* src/MG_solver_CPU.cpp:566-571 */
void orc_doGridAddition(int N, double *U1, const double *U2)
{
    const size_t n = (size_t)N * N;
#pragma omp parallel for
    for (size_t i = 0; i < n; ++i) U1[i] = U1[i] + U2[i];
}

/* src/MG_solver_CPU.cpp:573-625.
 * Both colour passes (:587-599) read only the copy made at the top of the step
 * (:581-585), so one step is one Jacobi sweep in correction form:
 *   U <- U + 0.25*(star(U_old) - 4*U_old - dx^2*F), rim untouched.
 * error (:607-622): sum1 and sum2 run over the SAME parity (first interior column
 * is 2 on even rows, 1 on odd rows, i.e. (row+col) even), so
 *   error = 2 * sum_{(row+col) even, interior} |inv*star - F| / N / N. */
void orc_doSmoothing(int N, double L, double *U, const double *F, int step, double *error)
{
    const double dx = L / (double)(N - 1);
    const double dx2 = square(dx);
    const size_t n = (size_t)N * N;
    double *prev = (double *)malloc(n * sizeof(double));

    for (int s = 0; s < step; ++s) {
        memcpy(prev, U, n * sizeof(double));
#pragma omp parallel for
        for (int r = 1; r < N - 1; ++r) {
            for (int c = 1; c < N - 1; ++c) {
                const size_t p = (size_t)r * N + c;
                U[p] = prev[p] + 0.25 * (star_minus4(prev, p, N) - dx2 * F[p]);
            }
        }
    }
    free(prev);

    const double inv = 1.0 / dx2;
    double half = 0.0;
#pragma omp parallel for reduction(+ : half)
    for (int r = 1; r < N - 1; ++r) {
        double row_sum = 0.0;
        for (int c = (r % 2 == 0) ? 2 : 1; c < N - 1; c += 2) {
            const size_t p = (size_t)r * N + c;
            row_sum += fabs(inv * star_minus4(U, p, N) - F[p]);
        }
        half += row_sum;
    }
    double e = half + half; /* sum1 + sum2, identical sums (:610 vs :617) */
    e = e / N / N;
    *error = e;
}

static int g_last_gs_iterations = 0;
int orc_lastExactSolverIterations(void) { return g_last_gs_iterations; }

/* src/MG_solver_CPU.cpp:952-1066 (red-black Gauss-Seidel, option 1).
 * The ieven/iodd tables (:973-990) enumerate, for every N, the points with
 * (col+row) even resp. odd (entries that alias onto col==N decode to a rim point
 * of the next row and are skipped, :1009); they are rebuilt here literally.
 * Update (:1020,:1043): U = 0.25*(left + right + up + down - h^2*F), in place.
 * err (:1051-1059) = sum_{interior}|residual| / (N-2)^2; stop when err <= target. */
static void gauss_seidel(int N, double L, double *U, const double *F, double target)
{
    const double h = L / (double)(N - 1);
    const double h2 = square(h);
    const int half = (N * N) / 2;
    int *first = (int *)malloc((size_t)(half > 0 ? half : 1) * sizeof(int));
    int *second = (int *)malloc((size_t)(half > 0 ? half : 1) * sizeof(int));
    double *R = (double *)malloc((size_t)N * N * sizeof(double));

    for (int i = 0; i < half; ++i) {
        int col = (2 * i) % N;
        const int row = ((2 * i) / N) % N;
        first[i] = col + ((col + row) % 2) + row * N;
        second[i] = col + ((col + row + 1) % 2) + row * N;
    }
    memset(U, 0, (size_t)N * N * sizeof(double));

    double err = target + 1.0;
    int iterations = 0;
    while (err > target) {
        for (int pass = 0; pass < 2; ++pass) {
            const int *tab = pass == 0 ? first : second;
#pragma omp parallel for
            for (int i = 0; i < half; ++i) {
                const int p = tab[i];
                const int col = p % N, row = p / N;
                if (on_rim(row, col, N)) continue;
                U[p] = 0.25 * (U[p - 1] + U[p + 1] + U[p + N] + U[p - N] - h2 * F[p]);
            }
        }
        ++iterations;
        orc_getResidual(N, L, U, F, R);
        double sum = 0.0;
#pragma omp parallel for reduction(+ : sum)
        for (int row = 1; row < N - 1; ++row) {
            double rs = 0.0;
            for (int col = 1; col < N - 1; ++col) rs = rs + fabs(R[col + N * row]);
            sum += rs;
        }
        err = sum / (double)((N - 2) * (N - 2));
    }
    g_last_gs_iterations = iterations;
    free(first);
    free(second);
    free(R);
}

/* src/MG_solver_CPU.cpp:627-638.  option 0 (dense inverse, :758-950) is out of the
 * hot-path scope; the oracle rejects it loudly. */
void orc_doExactSolver(int N, double L, double *U, const double *F, double target_error, int option)
{
    if (option == 1) {
        gauss_seidel(N, L, U, F, target_error);
    } else {
        g_last_gs_iterations = -1;
    }
}

/* src/MG_solver_CPU.cpp:647-648, 661-666: spacing is 1/(N-1) and 1/(M-1), L is
 * ignored; the same 1-D table serves columns and rows. */
void orc_restrictionTable(int N, int M, int *lo, double *w)
{
    const double h_f = 1.0 / (double)(N - 1);
    const double h_c = 1.0 / (double)(M - 1);
    for (int i = 0; i < M; ++i) {
        lo[i] = (int)floor((double)i * h_c / h_f);
        w[i] = fmod((double)i * h_c, h_f) / h_f;
    }
}

/* src/MG_solver_CPU.cpp:640-680: coarse array zeroed (:651), interior coarse points
 * sample the fine grid bilinearly (:674-676), products and sums left to right. */
void orc_doRestriction(int N, const double *U_f, int M, double *U_c)
{
    int *lo = (int *)malloc((size_t)M * sizeof(int));
    double *w = (double *)malloc((size_t)M * sizeof(double));
    orc_restrictionTable(N, M, lo, w);
    memset(U_c, 0, (size_t)M * M * sizeof(double));
#pragma omp parallel for
    for (int rc = 1; rc < M - 1; ++rc) {
        const double c = w[rc], d = 1.0 - c;
        for (int cc = 1; cc < M - 1; ++cc) {
            const double a = w[cc], b = 1.0 - a;
            const size_t f = (size_t)lo[cc] + (size_t)lo[rc] * N;
            U_c[cc + (size_t)rc * M] =
                b * d * U_f[f] + a * d * U_f[f + 1] + c * b * U_f[f + N] + a * c * U_f[f + N + 1];
        }
    }
    free(lo);
    free(w);
}

/* src/MG_solver_CPU.cpp:682-724.  N coarse, M fine, L = 1 (:683).  Every coarse
 * cell (i,j) writes the fine points k in [ceil(i*ratio), ceil((i+1)*ratio)) x l in
 * [ceil(j*ratio), ceil((j+1)*ratio)) (:697-698) with the bilinear expression of :700;
 * the cell holding l == M-2 also writes column M-1 with f_x = L (:701-704), and the
 * cell holding k == M-2 also writes row M-1 with f_y = (M-1)*f_dx and then leaves
 * its k loop (:706-718). */
static inline double prolong_value(double c1, double c2, double c3, double c4, double c1x,
                                   double c2x, double c1y, double c3y, double f_x, double f_y,
                                   double c_dx)
{
    return ((c1 * (c2x - f_x) + c2 * (f_x - c1x)) * (c3y - f_y) +
            (c3 * (c2x - f_x) + c4 * (f_x - c1x)) * (f_y - c1y)) /
           c_dx / c_dx;
}

void orc_doProlongation(int N, const double *U_c, int M, double *U_f)
{
    const double L = 1.0;
    const double c_dx = L / (double)(N - 1), f_dx = L / (double)(M - 1);
    const double ratio = c_dx / f_dx;
#pragma omp parallel for
    for (int i = 0; i < N - 1; ++i) {
        for (int j = 0; j < N - 1; ++j) {
            const double c1x = j * c_dx, c1y = i * c_dx;
            const double c2x = c1x + c_dx, c3y = c1y + c_dx;
            const double c1 = U_c[(size_t)i * N + j], c2 = U_c[(size_t)i * N + j + 1];
            const double c3 = U_c[(size_t)(i + 1) * N + j], c4 = U_c[(size_t)(i + 1) * N + j + 1];
            const int l_begin = (int)ceil(j * ratio);
            const double l_end = ceil((j + 1) * ratio);
            const double k_end = ceil((i + 1) * ratio);
            for (int k = (int)ceil(i * ratio); k < k_end; ++k) {
                const int rows_here = (k == M - 2) ? 2 : 1; /* row k, then forced row M-1 */
                for (int pass = 0; pass < rows_here; ++pass) {
                    const int kk = pass == 0 ? k : M - 1;
                    const double f_y = kk * f_dx;
                    for (int l = l_begin; l < l_end; ++l) {
                        U_f[(size_t)kk * M + l] = prolong_value(c1, c2, c3, c4, c1x, c2x, c1y, c3y,
                                                                l * f_dx, f_y, c_dx);
                        if (l == M - 2)
                            U_f[(size_t)kk * M + M - 1] = prolong_value(c1, c2, c3, c4, c1x, c2x, c1y,
                                                                        c3y, L, f_y, c_dx);
                    }
                }
                if (k == M - 2) break; /* the reference sets k = M-1 here and its loop ends */
            }
        }
    }
}

void orc_prolongationOwner(int N, int M, int *owner)
{
    const double c_dx = 1.0 / (double)(N - 1), f_dx = 1.0 / (double)(M - 1);
    const double ratio = c_dx / f_dx;
    for (int k = 0; k < M; ++k) owner[k] = -1;
    for (int i = 0; i < N - 1; ++i) {
        const double k_end = ceil((i + 1) * ratio);
        for (int k = (int)ceil(i * ratio); k < k_end; ++k) {
            if (k >= 0 && k < M) owner[k] = i;
            if (k == M - 2) { owner[M - 1] = i; break; }
        }
    }
}
