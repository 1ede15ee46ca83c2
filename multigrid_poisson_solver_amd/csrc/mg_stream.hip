// mg_stream.hip -- the fp64 instantiation of the temporally blocked wave-streaming smoother
// (kernel source and design notes: mg_stream_impl.h).
#define MG_REAL double
#define MG_REAL_NS f64
#include "mg_stream_impl.h"

namespace mg {
namespace k {

int stream_max_steps() { return f64::MAX_S; }
bool stream_supported(int N) { return N >= 3; }
bool stream_fusable(int N) { return N >= 4 && N % 2 == 0; }
bool stream_recompute_supported(int pre, int steps) { return f64::recompute_instantiated(pre, steps); }

void jacobi_stream(hipStream_t s, int N, double dx2, double inv, const double *in, const double *F, double *out,
                   int steps, double *err_out, double *D_out, int d_sign, const double *coarse, int Nc,
                   const ProlongTable *pt, double *Fc, int M, const RestrictTable *rt, const RowWindow *fine_w,
                   const RowWindow *coarse_w, const RowWindow *fc_w, int pre, bool no_out, const NodeBatch *batch)
{
    f64::StreamTables tb;
    if (coarse) {
        tb.p_orow = pt->owner_row;
        tb.p_ocol = pt->owner_col;
        tb.p_rhi = pt->row_hi;
        tb.p_rlo = pt->row_lo;
        tb.p_chi = pt->col_hi;
        tb.p_clo = pt->col_lo;
        tb.c_dx = pt->c_dx;
        tb.c_dx_rcp = 1.0 / pt->c_dx;  // IEEE division on the host: correctly rounded
    }
    if (Fc) {
        tb.r_inv = rt->inv;
        tb.r_w = rt->w;
        tb.r_wf = rt->inv_w;
    }
    f64::run(s, N, dx2, inv, in, F, out, steps, err_out, D_out, d_sign, coarse, Nc, Fc, M, tb, fine_w, coarse_w, fc_w, nullptr, pre, no_out, batch);
}

}  // namespace k
}  // namespace mg
