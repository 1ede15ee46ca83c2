// mg_tile.hip -- the fp64 instantiation of the register-tile fused nodes of the small levels
// (kernel source and design notes: mg_tile_impl.h).
#define MG_REAL double
#define MG_REAL_NS f64
#include "mg_tile_impl.h"

namespace mg {
namespace k {

// levels the fused driver hands to the tile kernel: MG_TILE_MIN_N <= N <= MG_TILE_MAX_N (defaults 65 and 1024: below, the
// coarse-tail kernel holds the levels in LDS; above, the arrays leave the caches and the streaming kernel's one pass
// over memory wins).  MG_TILE_MAX_N=0 switches the kernel off.
bool tile_wanted(int N)
{
    static const int lo = [] { const char *e = getenv("MG_TILE_MIN_N"); return e ? atoi(e) : 65; }();
    static const int hi = [] { const char *e = getenv("MG_TILE_MAX_N"); return e ? atoi(e) : 1024; }();
    return N >= lo && N <= hi && N >= 16;
}
// a slab of a distributed level is a small launch whatever the level's N: the streaming kernel's launch floor (14-15 us for
// the 256-row slabs of a 2048 level) is what the tile kernel removes
bool tile_wanted_slab(int N)
{
    static const int hi = [] { const char *e = getenv("MG_TILE_SLAB_MAX_N"); return e ? atoi(e) : 2048; }();
    static const int lo = [] { const char *e = getenv("MG_TILE_MIN_N"); return e ? atoi(e) : 65; }();
    return N >= lo && N <= hi && N >= 16;
}
int tile_max_steps() { return f64::tile::MAX_S; }

void jacobi_tile(hipStream_t s, int N, double dx2, double inv, const double *in, const double *F, double *out, int steps,
                 double *err_out, int d_sign, const double *coarse, int Nc, const ProlongTable *pt, double *Fc, int M,
                 const RestrictTable *rt, bool no_out, const RowWindow *fine_w, const RowWindow *coarse_w, const RowWindow *fc_w,
                 const NodeBatch *batch)
{
    f64::tile::Tables tb;
    if (coarse) {
        tb.p_orow = pt->owner_row;
        tb.p_ocol = pt->owner_col;
        tb.p_rhi = pt->row_hi;
        tb.p_rlo = pt->row_lo;
        tb.p_chi = pt->col_hi;
        tb.p_clo = pt->col_lo;
        tb.c_dx = pt->c_dx;
        tb.c_dx_rcp = 1.0 / pt->c_dx;  // IEEE division on the host: correctly rounded
        tb.p_closed = pt->closed_form;
    }
    if (Fc) {
        tb.r_inv = rt->inv;
        tb.r_w = rt->w;
        tb.r_wf = rt->inv_w;
    }
    f64::tile::run(s, N, dx2, inv, in, F, out, steps, err_out, d_sign, coarse, Nc, Fc, M, tb, no_out, fine_w, coarse_w, fc_w, batch);
}

}  // namespace k
}  // namespace mg
