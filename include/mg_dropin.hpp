// mg_dropin.hpp -- lets the reference's own driver compile against libmgpoisson.so.
//
// The reference declares its operators as C++ free functions
// (src/MG_solver_CPU.cpp:16-30; GPU twins with a _GPU suffix, src/MG_solver_GPU.cu:27-41)
// and defines them in the same file as main().  A maintainer who wants the MI355X path
// deletes those definitions (or renames main's calls), includes this header instead, and
// swaps the three malloc() calls of ListNode (src/linkedlist.cpp:9-11) and the tempU
// malloc (src/MG_solver_CPU.cpp:353) for mg_alloc() -- see INTEGRATION.md for the full
// diff.  Argument lists, order and meaning are identical; U/F/D are device arrays.
#pragma once
#include "mg_hip.h"

inline void getSource(int N, double L, double* F, double min_x, double min_y) { mg_getSource(N, L, F, min_x, min_y); }
inline void getAnalytic(int N, double L, double* U, double min_x, double min_y) { mg_getAnalytic(N, L, U, min_x, min_y); }
inline void getResidual(int N, double L, double* U, double* F, double* D) { mg_getResidual(N, L, U, F, D); }
inline void doGridAddition(int N, double* U1, double* U2) { mg_doGridAddition(N, U1, U2); }
inline void doSmoothing(int N, double L, double* U, double* F, int step, double* error) { mg_doSmoothing(N, L, U, F, step, error); }
inline void doExactSolver(int N, double L, double* U, double* F, double target_error, int option) { mg_doExactSolver(N, L, U, F, target_error, option); }
inline void doRestriction(int N, double* U_f, int M, double* U_c) { mg_doRestriction(N, U_f, M, U_c); }
inline void doProlongation(int N, double* U_c, int M, double* U_f) { mg_doProlongation(N, U_c, M, U_f); }
inline void doPrint2File(int N, double* U, char* file_name) { mg_print2File(N, U, file_name); }

// the same names with the reference's GPU-build suffix (src/MG_solver_GPU.cu:34-39)
inline void getSource_GPU(int N, double L, double* F, double min_x, double min_y) { mg_getSource(N, L, F, min_x, min_y); }
inline void getResidual_GPU(int N, double L, double* U, double* F, double* D) { mg_getResidual(N, L, U, F, D); }
inline void doGridAddition_GPU(int N, double* U1, double* U2) { mg_doGridAddition(N, U1, U2); }
inline void doSmoothing_GPU(int N, double L, double* U, double* F, int step, double* error) { mg_doSmoothing(N, L, U, F, step, error); }
inline void doExactSolver_GPU(int N, double L, double* U, double* F, double target_error, int option) { mg_doExactSolver(N, L, U, F, target_error, option); }
inline void doRestriction_GPU(int N, double* U_f, int M, double* U_c) { mg_doRestriction(N, U_f, M, U_c); }
inline void doProlongation_GPU(int N, double* U_c, int M, double* U_f) { mg_doProlongation(N, U_c, M, U_f); }
