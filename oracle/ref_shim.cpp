// ref_shim.cpp -- TEST INFRASTRUCTURE.  Compiles the reference's own CPU program
// from the sources where they lie (-I/root/reference/src; nothing is copied into this
// repository) and re-exports its operators with C linkage so tests/ and bench.py's
// cpu_baseline leg can call the real reference code.  The output goes to oracle/_ref/
// only (git-ignored, but shipped to the GPU box as a built artefact).
//
// The reference translation unit defines main(); it is renamed while the file is
// included so the operators become callable from a shared library.
#define main mg_reference_program_main
#include "MG_solver_CPU.cpp"  // resolved through -I to /root/reference/src
#undef main

extern "C" {
void ref_getSource(int N, double L, double* F, double mx, double my) { getSource(N, L, F, mx, my); }
void ref_getAnalytic(int N, double L, double* U, double mx, double my) { getAnalytic(N, L, U, mx, my); }
void ref_getResidual(int N, double L, double* U, double* F, double* D) { getResidual(N, L, U, F, D); }
void ref_doGridAddition(int N, double* U1, double* U2) { doGridAddition(N, U1, U2); }
void ref_doSmoothing(int N, double L, double* U, double* F, int step, double* error) { doSmoothing(N, L, U, F, step, error); }
void ref_doExactSolver(int N, double L, double* U, double* F, double tol, int option) { doExactSolver(N, L, U, F, tol, option); }
void ref_doRestriction(int N, double* U_f, int M, double* U_c) { doRestriction(N, U_f, M, U_c); }
void ref_doProlongation(int N, double* U_c, int M, double* U_f) { doProlongation(N, U_c, M, U_f); }
void ref_setThreads(int n) { omp_set_num_threads(n); }
int  ref_program(int argc, char** argv) { return mg_reference_program_main(argc, argv); }
}
