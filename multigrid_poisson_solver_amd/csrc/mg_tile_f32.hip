// mg_tile_f32.hip -- the fp32 instantiation of the register-tile fused nodes of the small levels (kernel source:
// mg_tile_impl.h): the small levels of the mixed-precision mode's fp32 cycle.  Weights are the fp64 host tables rounded
// to fp32, norms are accumulated in fp64 -- as in mg_stream_f32.hip.
#define MG_REAL float
#define MG_REAL_NS f32
#include "mg_tile_impl.h"

namespace mg {
namespace k {

void jacobi_tile_f32(hipStream_t s, int N, float dx2, float inv, const float *in, const float *F, float *out, int steps,
                     double *err_out, int d_sign, const float *coarse, int Nc, const ProlongTable *pt, float *Fc, int M,
                     const RestrictTable *rt, bool no_out, const RowWindow *fine_w, const RowWindow *coarse_w, const RowWindow *fc_w)
{
    f32::tile::Tables tb;
    if (coarse) {
        tb.p_orow = pt->owner_row;
        tb.p_ocol = pt->owner_col;
        tb.p_rhi = pt->row_hi_f;
        tb.p_rlo = pt->row_lo_f;
        tb.p_chi = pt->col_hi_f;
        tb.p_clo = pt->col_lo_f;
        tb.c_dx = (float)pt->c_dx;
        tb.c_dx_rcp = 1.0f / tb.c_dx;  // IEEE fp32 division on the host: correctly rounded
        tb.p_closed = pt->closed_form;
    }
    if (Fc) {
        tb.r_inv = rt->inv;
        tb.r_w = rt->w_f;
        tb.r_wf = rt->inv_w_f;
    }
    f32::tile::run(s, N, dx2, inv, in, F, out, steps, err_out, d_sign, coarse, Nc, Fc, M, tb, no_out, fine_w, coarse_w, fc_w);
}

}  // namespace k
}  // namespace mg
