// mg_stream_f32.hip -- the fp32 instantiation of the temporally blocked wave-streaming smoother
// (kernel source: mg_stream_impl.h): the smoother of the mixed-precision mode (SURVEY.md section
// 8f-2, BASELINE.json config 5: fp32 smoothing, fp64 residual and correction).  Every array and
// every arithmetic operation is fp32 (8 B per lane and row); the transfer weights are the fp64
// host tables rounded to fp32; norms are accumulated in fp64.
#define MG_REAL float
#define MG_REAL_NS f32
#include "mg_stream_impl.h"

namespace mg {
namespace k {

void jacobi_stream_f32(hipStream_t s, int N, float dx2, float inv, const float *in, const float *F, float *out, int steps,
                       double *err_out, const float *coarse, int Nc, const ProlongTable *pt, float *Fc, int M,
                       const RestrictTable *rt, const RowWindow *fine_w, const RowWindow *coarse_w, const RowWindow *fc_w,
                       double *out_wide, float *D_out, int d_sign, int pre, bool no_out)
{
    f32::StreamTables tb;
    if (coarse) {
        tb.p_orow = pt->owner_row;
        tb.p_ocol = pt->owner_col;
        tb.p_rhi = pt->row_hi_f;
        tb.p_rlo = pt->row_lo_f;
        tb.p_chi = pt->col_hi_f;
        tb.p_clo = pt->col_lo_f;
        tb.c_dx = (float)pt->c_dx;
        tb.c_dx_rcp = 1.0f / tb.c_dx;  // IEEE fp32 division on the host: correctly rounded
        tb.p_cols4_ok = pt->fusable4;
    }
    if (Fc) {
        tb.r_inv = rt->inv;
        tb.r_w = rt->w_f;
        tb.r_wf = rt->inv_w_f;
    }
    f32::run(s, N, dx2, inv, in, F, out, steps, err_out, D_out, d_sign, coarse, Nc, Fc, M, tb, fine_w, coarse_w, fc_w, out_wide, pre, no_out);
}

}  // namespace k
}  // namespace mg
