// mg_abi.cpp -- the extern "C" boundary (include/mg_hip.h): context, caching pool,
// and the operator entry points that launch the kernels of mg_kernels.hip /
// mg_stream.hip.  There is NO CPU fallback anywhere in this file: without a HIP
// device every entry point fails loudly.
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>

#include "mg_internal.h"

namespace mg {

Context &ctx()
{
    static Context c;
    return c;
}

void fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    Context &c = ctx();
    c.last_error = code;
    c.last_error_text = buf;
    fprintf(stderr, "[ ERROR ]: %s\n", buf);
    if (c.abort_on_error) exit(1);  // the reference's convention: printf + exit(1)
}

bool hip_ok(hipError_t e, const char *what, const char *file, int line)
{
    if (e == hipSuccess) return true;
    fail(MG_ERR_HIP, "HIP call failed: %s -> %s (%s:%d)", what, hipGetErrorString(e), file, line);
    return false;
}

bool require_ready(const char *who)
{
    if (ctx().ready) return true;
    fail(MG_ERR_NOT_INIT, "%s: mg_init() has not been called (or failed); there is no CPU fallback", who);
    return false;
}

// ------------------------------------------------------------------ pool
void *Pool::get(size_t bytes)
{
    if (bytes == 0) bytes = 8;
    bytes = (bytes + 255) & ~(size_t)255;
    auto it = free_.find(bytes);
    if (it != free_.end()) {
        void *p = it->second;
        free_.erase(it);
        live_[p] = bytes;
        return p;
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {  // give cached blocks back and retry once
        (void)hipGetLastError();
        trim();
        if (!MG_HIP(hipMalloc(&p, bytes))) return nullptr;
    }
    held_ += bytes;
    live_[p] = bytes;
    return p;
}

void Pool::put(void *p)
{
    if (!p) return;
    auto it = live_.find(p);
    if (it == live_.end()) {
        fail(MG_ERR_ARG, "mg_free: pointer %p was not returned by mg_alloc", p);
        return;
    }
    if (park_) parked_.emplace_back(it->second, p);
    else free_.emplace(it->second, p);
    live_.erase(it);
}

void Pool::release_parked()
{
    for (auto &kv : parked_) free_.emplace(kv.first, kv.second);
    parked_.clear();
}

void Pool::trim()
{
    for (auto &kv : free_) {
        (void)hipFree(kv.second);
        held_ -= kv.first;
    }
    free_.clear();
}

Pool &scratch_pool()
{
    Context &c = ctx();
    return c.active_pool ? *c.active_pool : c.pool;
}

double *partials(size_t n)
{
    Context &c = ctx();
    if (n > c.partials_cap) {
        // grown outside of any hot loop: the first call at a given size allocates
        if (c.partials) c.retired.push_back(c.partials);  // a captured graph may still point at it
        size_t cap = n < 4096 ? 4096 : n;
        if (!MG_HIP(hipMalloc((void **)&c.partials, cap * sizeof(double)))) return nullptr;
        c.partials_cap = cap;
    }
    return c.partials;
}

// ------------------------------------------------------------------ live timing
static bool stream_is_capturing(hipStream_t s)
{
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return st != hipStreamCaptureStatusNone;
}

ProfScope::ProfScope(const char *name, int N, double algo_bytes, hipStream_t stream)
{
    Context &c = ctx();
    on = stream ? stream : c.stream;
    if (!c.profiling || N < c.profile_min_N || stream_is_capturing(on)) return;
    if (c.profile_every > 1 && c.profile_window % c.profile_every != 1 % c.profile_every) return;  // (windows count from 1)
    hipEvent_t e[2];
    for (int i = 0; i < 2; ++i) {
        if (!c.event_pool.empty()) {
            e[i] = c.event_pool.back();
            c.event_pool.pop_back();
        } else if (hipEventCreate(&e[i]) != hipSuccess) {
            return;
        }
    }
    Context::ProfRec r{name, N, algo_bytes, e[0], e[1]};
    c.prof.push_back(r);
    slot = (int)c.prof.size() - 1;
    (void)hipEventRecord(e[0], on);
}

ProfScope::~ProfScope()
{
    if (slot < 0) return;
    Context &c = ctx();
    (void)hipEventRecord(c.prof[slot].e1, on);
}

double *norm_partials(size_t n)
{
    Context &c = ctx();
    if (!c.defer_norms) return partials(n);
    c.norm_window_total += n;
    if (c.norm_arena_used + n > c.norm_arena_cap) {
        // grow; what is pending keeps pointing into the old arena, which stays alive
        size_t cap = c.norm_arena_cap ? c.norm_arena_cap * 2 : (size_t)1 << 16;
        while (cap < n) cap *= 2;
        if (c.norm_arena) c.retired.push_back(c.norm_arena);
        c.norm_arena = nullptr;
        if (!MG_HIP(hipMalloc((void **)&c.norm_arena, cap * sizeof(double)))) return nullptr;
        c.norm_arena_cap = cap;
        c.norm_arena_used = 0;
    }
    double *p = c.norm_arena + c.norm_arena_used;
    c.norm_arena_used += n;
    return p;
}

void norm_finish(hipStream_t s, const double *part, size_t n, int N, double *out)
{
    Context &c = ctx();
    if (!c.defer_norms) {
        k::finish_smoothing_error(s, part, n, N, out);
        return;
    }
    c.pending_norms.push_back(Context::PendingNorm{part, (int)n, N, out});
    if ((int)c.pending_norms.size() >= k::MAX_NORMS_PER_FLUSH && !c.norms_at_window_end) flush_norms();
}

void flush_norms()
{
    Context &c = ctx();
    if (c.pending_norms.empty()) return;
    k::NormBatch b;
    int count = 0;
    for (const auto &pn : c.pending_norms) {
        b.part[count] = pn.part;
        b.out[count] = pn.out;
        b.n[count] = pn.n;
        b.N[count] = pn.N;
        if (++count == k::MAX_NORMS_PER_FLUSH) {   // (only a window that held its norms back to the end has more than one batch)
            k::finish_smoothing_errors(c.stream, b, count);
            count = 0;
        }
    }
    if (count) k::finish_smoothing_errors(c.stream, b, count);
    c.pending_norms.clear();
    c.norm_arena_used = 0;  // stream order: later partials are written after this reduction ran
    // An arena that grew in mid-window holds only what came after its last growth: size it for ALL partials between two
    // flushes now, so that the same window run again -- possibly under stream capture, where hipMalloc is not allowed --
    // finds room.  (The old arena stays alive: the reduction just enqueued reads it.)
    if (c.norm_window_total > c.norm_arena_cap && !stream_is_capturing(c.stream)) {
        size_t cap = c.norm_arena_cap ? c.norm_arena_cap : (size_t)1 << 16;
        while (cap < c.norm_window_total) cap *= 2;
        double *bigger = nullptr;
        if (hipMalloc((void **)&bigger, cap * sizeof(double)) == hipSuccess) {
            if (c.norm_arena) c.retired.push_back(c.norm_arena);
            c.norm_arena = bigger;
            c.norm_arena_cap = cap;
        } else {
            (void)hipGetLastError();
        }
    }
    c.norm_window_total = 0;
}

namespace {

struct Scalars {  // slots inside ctx().scalars
    enum { SMOOTH_ERR = 0, ANALYTIC_ERR = 1, CHECKSUM = 2 /* 2 x u64 */ };
};

bool grid_args_ok(const char *who, int N)
{
    if (N < 3) {
        fail(MG_ERR_ARG, "%s: grid size N=%d is too small (need N >= 3)", who, N);
        return false;
    }
    if (N > 46340) {  // N*N must fit the reference's int indexing
        fail(MG_ERR_ARG, "%s: grid size N=%d overflows int indexing", who, N);
        return false;
    }
    return true;
}

// h^2 as the reference forms it: pow(dx, 2) with dx = L/(double)(N-1)
// (src/MG_solver_CPU.cpp:555,574,954), taken as dx*dx -- what every optimising build of
// the reference computes; glibc's pow() differs by one ulp for a few grid sizes (2948,
// 3504, ...: none a power of two), see oracle/mg_oracle.c:square.
inline double spacing_sq(int N, double L)
{
    const double dx = L / (double)(N - 1);
    return dx * dx;
}

// fused transfer stages (streaming smoother only): level 0 = U_in + P(coarse) for the first
// launch, restriction of the signed residual into Fc for the last launch
struct Fusion {
    const double *coarse = nullptr;
    int Nc = 0;
    const ProlongTable *pt = nullptr;
    double *Fc = nullptr;
    int M = 0;
    const RestrictTable *rt = nullptr;
    int pre = 0;          // `1` node: recompute the pre-smoothed field (pre sweeps from zero) instead of reading U_in
    bool no_out = false;  // `-1` node: do not store the smoothed field
};

void smooth_pp(int N, double L, const double *U_in, double *U_out, double *F, int step, double *error_dev,
               double *D_out, int d_sign, const Fusion &fu = Fusion())
{
    Context &c = ctx();
    hipStream_t s = c.stream;
    const double dx2 = spacing_sq(N, L);
    const double inv = 1.0 / dx2;
    const size_t n = (size_t)N * N;

    if (step <= 0 && c.trace) {
        c.trace_failed = true;
        return;
    }
    if (step <= 0) {  // no sweep: U_out = U_in
        if (U_in) (void)hipMemcpyAsync(U_out, U_in, n * sizeof(double), hipMemcpyDeviceToDevice, s);
        else (void)hipMemsetAsync(U_out, 0, n * sizeof(double), s);
        if (error_dev) k::smoothing_error(s, N, inv, U_out, F, error_dev);
        if (D_out) k::residual(s, N, inv, U_out, F, D_out, d_sign);
        return;
    }

    const bool stream = c.smoother != SMOOTHER_SIMPLE && k::stream_supported(N);
    if (c.trace) {
        // the cycle driver's dataflow trace: describe the launch instead of enqueueing it.  Only a node that is ONE fused
        // launch of the streaming or the tile kernel can be batched with its peers
        const int smax1 = stream ? k::stream_max_steps() : 1;
        if (!stream || step > smax1 || D_out || !error_dev || (fu.coarse && fu.Fc)) {
            c.trace_failed = true;
            return;
        }
        NodeOp op;
        // (levels between the batched schedule's coarse tail and the tile kernel's usual range: the tile kernel as well)
        const bool as_tile = c.smoother == SMOOTHER_STREAM && fu.pre == 0 && (k::tile_wanted(N) || (N >= 16 && N <= k::TAIL_MAX_N)) && step <= k::tile_max_steps();
        op.kind = as_tile ? 1 : 0;
        op.N = N;
        op.L = L;
        op.src = U_in;
        op.F = F;
        op.dst = U_out;
        op.take = step;
        op.err = error_dev;
        op.d_sign = d_sign;
        op.coarse = fu.coarse;
        op.Nc = fu.Nc;
        op.Fc = fu.Fc;
        op.M = fu.M;
        op.pre = fu.pre;
        op.no_out = fu.no_out;
        const bool pro = fu.coarse != nullptr, rst = fu.Fc != nullptr;
        char pre_tag[8] = "";
        if (fu.pre) snprintf(pre_tag, sizeof pre_tag, ",pre%d", fu.pre);
        snprintf(op.name, sizeof op.name, "%s<%d%s%s%s%s%s>", as_tile ? "jacobi_tile" : "jacobi_stream", step, (U_in || fu.pre) ? "" : ",zero",
                 pro ? ",prolong" : "", rst ? ",res,restrict" : "", fu.no_out ? ",noU" : "", pre_tag);
        op.bytes = (double)n * (24.0 * (step + fu.pre) + ((U_in || fu.pre) ? 0.0 : 8.0) + (rst ? 24.0 : 0.0));
        if (rst) op.bytes += 8.0 * n + 8.0 * fu.M * fu.M;
        if (pro) op.bytes += 16.0 * n + 8.0 * fu.Nc * fu.Nc;
        c.trace->push_back(op);
        return;
    }
    // sweeps per launch: the streaming kernel advances up to stream_max_steps() time
    // levels per pass over HBM, the simple kernel one.  The launch count is made odd so
    // that, ping-ponging between U_out and one partner buffer, the last launch lands in
    // U_out and launch 0 never writes the buffer it reads.
    const int smax = stream ? k::stream_max_steps() : 1;
    int launches = (step + smax - 1) / smax;
    if (launches % 2 == 0 && step > launches) launches += 1;
    const bool needs_copy = (launches % 2 == 0);  // only smax == 1 with an even step
    double *partner = const_cast<double *>(U_in);
    bool own_partner = false;
    if ((launches > 1 && !partner) || needs_copy) {
        partner = (double *)scratch_pool().get(n * sizeof(double));
        own_partner = true;
        if (!partner) return;
    }
    const double *src = U_in;
    int left = step;
    for (int i = 0; i < launches; ++i) {
        const int take = (left + (launches - i) - 1) / (launches - i);
        const bool last = (i == launches - 1);
        // odd count: dst alternates so that the last is U_out.  even count (own partner):
        // start in the partner, the last launch then lands in U_out as well.
        double *dst = ((launches - 1 - i) % 2 == 0) ? U_out : partner;
        const bool plain_single = stream && c.smoother != SMOOTHER_STREAM_ONLY && take == 1 && src && !(i == 0 && fu.coarse) && !(last && (D_out || fu.Fc || error_dev)) &&
                                  N >= 2048 && N % 2 == 0;
        // small levels: the whole node as one launch of the register-tile kernel (mg_tile_impl.h) -- the streaming kernel's
        // march down a column strip is pure latency there.  Same bits.  Not for a stored residual (the unfused driver) and
        // not for the recomputing pair (large levels only).
        const bool tile = stream && c.smoother == SMOOTHER_STREAM && launches == 1 && !D_out && fu.pre == 0 && k::tile_wanted(N) &&
                          take <= k::tile_max_steps();
        if (tile) {
            const bool pro = fu.coarse != nullptr, rst = fu.Fc != nullptr;
            char name[48];
            snprintf(name, sizeof name, "jacobi_tile<%d%s%s%s%s>", take, src ? "" : ",zero", pro ? ",prolong" : "", rst ? ",res,restrict" : "",
                     fu.no_out ? ",noU" : "");
            double bytes = (double)n * (24.0 * take + (src ? 0.0 : 8.0) + (rst ? 24.0 : 0.0));
            if (rst) bytes += 8.0 * n + 8.0 * fu.M * fu.M;
            if (pro) bytes += 16.0 * n + 8.0 * fu.Nc * fu.Nc;
            ProfScope ps(name, N, bytes);
            k::jacobi_tile(s, N, dx2, inv, src, F, dst, take, error_dev, d_sign, fu.coarse, fu.Nc, fu.pt, fu.Fc, fu.M, fu.rt, fu.no_out);
        } else if (plain_single) {
            // one bare sweep of a large grid: the one-row-per-block pair kernel is the faster of the two
            // (5.1 vs 4.5 TB/s at N = 8192; same bits) -- there is nothing to fuse and no row history to amortise
            ProfScope ps("jacobi_pair", N, (double)n * 24.0);
            k::jacobi_simple(s, N, dx2, src, F, dst);
        } else if (stream) {
            // algorithmic bytes: 24 B per sweep and point, + 8 for a folded
            // zero-fill, + 24 for a folded residual, + 8n + 8m for a folded restriction,
            // + 8m + 16n for a folded prolongation+addition; the fused error costs nothing
            const bool pro = (i == 0 && fu.coarse), res = last && (D_out || fu.Fc), rst = last && fu.Fc;
            char name[48], pre_tag[8] = "";
            if (fu.pre) snprintf(pre_tag, sizeof pre_tag, ",pre%d", fu.pre);
            snprintf(name, sizeof name, "jacobi_stream<%d%s%s%s%s%s%s>", take, (src || fu.pre) ? "" : ",zero", pro ? ",prolong" : "",
                     res ? ",res" : "", rst ? ",restrict" : "", fu.no_out ? ",noU" : "", pre_tag);
            double bytes = (double)n * (24.0 * (take + fu.pre) + ((src || fu.pre) ? 0.0 : 8.0) + (res ? 24.0 : 0.0));
            if (rst) bytes += 8.0 * n + 8.0 * fu.M * fu.M;
            if (pro) bytes += 16.0 * n + 8.0 * fu.Nc * fu.Nc;
            ProfScope ps(name, N, bytes);
            k::jacobi_stream(s, N, dx2, inv, src, F, dst, take, last ? error_dev : nullptr, last ? D_out : nullptr,
                             d_sign, pro ? fu.coarse : nullptr, fu.Nc, fu.pt, rst ? fu.Fc : nullptr, fu.M, fu.rt, nullptr, nullptr, nullptr,
                             fu.pre, fu.no_out);
        } else {
            // (k_jacobi_pair on even N, k_jacobi_simple on odd N: the launcher picks)
            ProfScope ps(src ? (N % 2 == 0 && N >= 512 ? "jacobi_pair" : "jacobi_simple") : "jacobi_simple<zero>", N, (double)n * (src ? 24.0 : 32.0));
            k::jacobi_simple(s, N, dx2, src, F, dst);
        }
        src = dst;
        left -= take;
    }
    if (own_partner) scratch_pool().put(partner);  // stream-ordered: later users queue behind us
    if (!stream) {
        if (error_dev) {
            ProfScope ps("smoothing_error", N, 0.0);
            k::smoothing_error(s, N, inv, U_out, F, error_dev);
        }
        if (D_out) {
            ProfScope ps("residual", N, (double)n * 24.0);
            k::residual(s, N, inv, U_out, F, D_out, d_sign);
        }
    }
}

}  // namespace

int recompute_min_n()
{
    static const int min_n = [] { const char *e = getenv("MG_RECOMPUTE_MIN_N"); return e ? atoi(e) : 4096; }();  // (measured: 2048 loses 4 us per level, 4096 gains 35, 8192 gains 130)
    // a batched schedule carries 2^l instances of level l in one launch: what decides there is the launch, not the grid
    if (ctx().recompute_min_override > 0) return ctx().recompute_min_override;
    return min_n;
}

bool recompute_available(int Nc, int N, int pre, int step)
{
    const int min_n = recompute_min_n();
    if (!k::stream_recompute_supported(pre, step) || N < min_n || ctx().smoother == SMOOTHER_SIMPLE || !k::stream_fusable(N)) return false;
    // BOTH nodes of the pair must be fused launches: the `1` node's prolongation Nc -> N (owners advance by at most one)
    // and the `-1` node's restriction N -> Nc (samples at least two fine columns apart: non-nested pairs such as
    // 4096 -> 3000 or con_N = 2's N -> N - 1 are not, and run store/re-read operator by operator)
    const ProlongTable &pt = prolong_table(Nc, N);
    if (!pt.owner_row || !pt.fusable) return false;
    const RestrictTable &rt = restrict_table(N, Nc);
    return rt.lo && rt.fusable;
}

void smooth_restrict_no_out(int N, double L, double *U_unused, double *F, int step, double *error_dev, int M, double *F_c)
{
    const RestrictTable &rt = restrict_table(N, M);
    if (!rt.lo || !rt.fusable || !k::stream_fusable(N) || step < 1 || step > k::stream_max_steps()) {
        fail(MG_ERR_UNSUPPORTED, "smooth_restrict_no_out: N=%d M=%d step=%d is not a fused `-1` node", N, M, step);
        return;
    }
    Fusion fu;
    fu.Fc = F_c;
    fu.M = M;
    fu.rt = &rt;
    fu.no_out = true;
    smooth_pp(N, L, nullptr, U_unused, F, step, error_dev, nullptr, -1, fu);
}

void prolong_smooth_recompute(int Nc, const double *U_c, int N, double L, double *U_out, double *F, int pre, int step, double *error_dev)
{
    if (!recompute_available(Nc, N, pre, step)) {
        fail(MG_ERR_UNSUPPORTED, "prolong_smooth_recompute: Nc=%d N=%d pre=%d step=%d", Nc, N, pre, step);
        return;
    }
    Fusion fu;
    fu.coarse = U_c;
    fu.Nc = Nc;
    fu.pt = &prolong_table(Nc, N);
    fu.pre = pre;
    smooth_pp(N, L, nullptr, U_out, F, step, error_dev, nullptr, +1, fu);
}

// One launch for n recorded instances of the same fused node (the independent visits of a level, mg_cycle.cpp); the
// name carries the batch size, the algorithmic bytes are those of all instances.
void replay_node(const NodeOp &op, const NodeBatch *batch)
{
    Context &c = ctx();
    const double dx2 = spacing_sq(op.N, op.L), inv = 1.0 / dx2;
    const ProlongTable *pt = op.coarse ? &prolong_table(op.Nc, op.N) : nullptr;
    const RestrictTable *rt = op.Fc ? &restrict_table(op.N, op.M) : nullptr;
    const int nb = batch ? batch->n : 1;
    char name[64];
    if (nb > 1) snprintf(name, sizeof name, "%s x%d", op.name, nb);
    else snprintf(name, sizeof name, "%s", op.name);
    ProfScope ps(name, op.N, op.bytes * nb);
    if (op.kind == 1)
        k::jacobi_tile(c.stream, op.N, dx2, inv, op.src, op.F, op.dst, op.take, op.err, op.d_sign, op.coarse, op.Nc, pt, op.Fc, op.M, rt, op.no_out,
                       nullptr, nullptr, nullptr, batch);
    else
        k::jacobi_stream(c.stream, op.N, dx2, inv, op.src, op.F, op.dst, op.take, op.err, nullptr, op.d_sign, op.coarse, op.Nc, pt, op.Fc, op.M, rt,
                         nullptr, nullptr, nullptr, op.pre, op.no_out, batch);
}

}  // namespace mg

using namespace mg;

extern "C" {

// ------------------------------------------------------------------ lifecycle
int mg_init(int device)
{
    Context &c = ctx();
    if (c.ready) return 0;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        fail(MG_ERR_HIP, "Cannot select GPU: no HIP device is visible (this engine has no CPU fallback)");
        return 1;
    }
    if (device < 0 || device >= count) {
        fail(MG_ERR_ARG, "Cannot select GPU %d (have %d)", device, count);
        return 1;
    }
    if (!MG_HIP(hipSetDevice(device))) return 1;
    hipDeviceProp_t prop;
    if (MG_HIP(hipGetDeviceProperties(&prop, device))) c.n_cu = prop.multiProcessorCount;
    if (!MG_HIP(hipStreamCreateWithFlags(&c.own_stream, hipStreamNonBlocking))) return 1;
    c.stream = c.own_stream;
    if (!MG_HIP(hipMalloc((void **)&c.scalars, 64 * sizeof(double)))) return 1;
    if (!MG_HIP(hipMalloc((void **)&c.gs_state, 4 * sizeof(int)))) return 1;
    if (!MG_HIP(hipMemset(c.gs_state, 0, 4 * sizeof(int)))) return 1;
    if (!MG_HIP(hipHostMalloc((void **)&c.host_scalars, 64 * sizeof(double), hipHostMallocDefault))) return 1;
    if (!MG_HIP(hipHostMalloc((void **)&c.host_ints, 4 * sizeof(int), hipHostMallocDefault))) return 1;
    c.device = device;
    c.ready = true;
    const char *sm = getenv("MG_SMOOTHER");
    if (sm && *sm) mg_set_smoother(sm);
    const char *src = getenv("MG_SOURCE");
    if (src && *src) mg_set_source(src);
    return 0;
}

void mg_finalize(void)
{
    Context &c = ctx();
    if (!c.ready) return;
    (void)hipStreamSynchronize(c.stream);
    for (auto &kv : c.rtab) {
        (void)hipFree(kv.second.lo);
        (void)hipFree(kv.second.w);
        (void)hipFree(kv.second.inv);
        (void)hipFree(kv.second.inv_w);
        (void)hipFree(kv.second.w_f);
        (void)hipFree(kv.second.inv_w_f);
    }
    for (auto &kv : c.ptab) {
        (void)hipFree(kv.second.owner_row);
        (void)hipFree(kv.second.owner_col);
        (void)hipFree(kv.second.row_hi);
        (void)hipFree(kv.second.row_lo);
        (void)hipFree(kv.second.col_hi);
        (void)hipFree(kv.second.col_lo);
        (void)hipFree(kv.second.row_hi_f);
        (void)hipFree(kv.second.row_lo_f);
        (void)hipFree(kv.second.col_hi_f);
        (void)hipFree(kv.second.col_lo_f);
    }
    c.rtab.clear();
    c.ptab.clear();
    c.pool.trim();
    (void)hipFree(c.partials);
    (void)hipFree(c.norm_arena);
    c.norm_arena = nullptr;
    c.norm_arena_cap = c.norm_arena_used = 0;
    c.pending_norms.clear();
    for (void *r : c.retired) (void)hipFree(r);
    c.retired.clear();
    c.partials = nullptr;
    c.partials_cap = 0;
    (void)hipFree(c.scalars);
    (void)hipFree(c.gs_state);
    (void)hipHostFree(c.host_scalars);
    (void)hipHostFree(c.host_ints);
    for (hipEvent_t e : c.event_pool) (void)hipEventDestroy(e);
    c.event_pool.clear();
    (void)hipStreamDestroy(c.own_stream);
    c.own_stream = c.stream = nullptr;
    c.ready = false;
}

void mg_set_stream(void *hip_stream)
{
    Context &c = ctx();
    if (!require_ready("mg_set_stream")) return;
    (void)hipStreamSynchronize(c.stream);
    c.stream = hip_stream ? (hipStream_t)hip_stream : c.own_stream;
}
void *mg_get_stream(void) { return (void *)ctx().stream; }
void mg_sync(void)
{
    if (!require_ready("mg_sync")) return;
    MG_HIP(hipStreamSynchronize(ctx().stream));
}
int mg_last_error(void) { return ctx().last_error; }
const char *mg_last_error_string(void) { return ctx().last_error_text.c_str(); }
void mg_clear_error(void)
{
    ctx().last_error = 0;
    ctx().last_error_text.clear();
}
void mg_set_abort_on_error(int on) { ctx().abort_on_error = on != 0; }
int mg_set_smoother(const char *name)
{
    if (name && strcmp(name, "stream") == 0) ctx().smoother = SMOOTHER_STREAM;
    else if (name && strcmp(name, "simple") == 0) ctx().smoother = SMOOTHER_SIMPLE;
    else if (name && strcmp(name, "stream_only") == 0) ctx().smoother = SMOOTHER_STREAM_ONLY;
    else {
        fail(MG_ERR_ARG, "mg_set_smoother: unknown smoother '%s' (stream|simple|stream_only)", name ? name : "(null)");
        return 1;
    }
    return 0;
}
}  // extern "C"
namespace mg {
bool source_selfcheck();
bool source_on_device();
}  // namespace mg
extern "C" {

int mg_set_source(const char *name)
{
    if (name && strcmp(name, "auto") == 0) ctx().source_mode = 0;
    else if (name && strcmp(name, "host") == 0) ctx().source_mode = 1;
    else if (name && strcmp(name, "device") == 0) ctx().source_mode = 2;
    else {
        fail(MG_ERR_ARG, "mg_set_source: unknown mode '%s' (auto|host|device)", name ? name : "(null)");
        return 1;
    }
    return 0;
}
const char *mg_source_mode(void)
{
    if (!require_ready("mg_source_mode")) return "host";
    return source_on_device() ? "device" : "host";
}
int mg_source_is_bit_identical(void)
{
    if (!require_ready("mg_source_is_bit_identical")) return 0;
    return source_selfcheck() ? 1 : 0;
}
const char *mg_version(void) { return "mgpoisson-hip 0.1 (gfx950)"; }

// ------------------------------------------------------------------ memory
double *mg_alloc(size_t n)
{
    if (!require_ready("mg_alloc")) return nullptr;
    return (double *)ctx().pool.get(n * sizeof(double));
}
void mg_free(double *p)
{
    if (!p || !require_ready("mg_free")) return;
    ctx().pool.put(p);
}
void mg_pool_trim(void)
{
    if (!ctx().ready) return;
    (void)hipStreamSynchronize(ctx().stream);
    ctx().pool.trim();
}
size_t mg_pool_bytes(void) { return ctx().pool.bytes_held(); }

void mg_upload(double *dev, const double *host, size_t n)
{
    if (!require_ready("mg_upload")) return;
    hipStream_t s = ctx().stream;
    if (MG_HIP(hipMemcpyAsync(dev, host, n * sizeof(double), hipMemcpyHostToDevice, s))) MG_HIP(hipStreamSynchronize(s));
}
void mg_download(double *host, const double *dev, size_t n)
{
    if (!require_ready("mg_download")) return;
    hipStream_t s = ctx().stream;
    if (MG_HIP(hipMemcpyAsync(host, dev, n * sizeof(double), hipMemcpyDeviceToHost, s))) MG_HIP(hipStreamSynchronize(s));
}
void mg_copy(double *dst, const double *src, size_t n)
{
    if (!require_ready("mg_copy")) return;
    MG_HIP(hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToDevice, ctx().stream));
}
void mg_fill_zero(double *dev, size_t n)
{
    if (!require_ready("mg_fill_zero")) return;
    MG_HIP(hipMemsetAsync(dev, 0, n * sizeof(double), ctx().stream));
}
void mg_negate(int N, double *D)
{
    if (!require_ready("mg_negate") || !grid_args_ok("mg_negate", N)) return;
    k::negate(ctx().stream, (size_t)N * N, D);
}

// ------------------------------------------------------------------ problem definition
namespace {
struct SourceJob {
    int N;
    double h, min_x, min_y;
    double *F;
};
// src/MG_solver_CPU.cpp:468-493 on the host: libm exp(), the same bits as the
// reference program produces on this machine
void source_rows(size_t r0, size_t r1, void *arg)
{
    const SourceJob &j = *(const SourceJob *)arg;
    const int N = j.N;
    for (size_t r = r0; r < r1; ++r) {
        for (int c = 0; c < N; ++c) {
            double v = 0.0;
            if (!(r == 0 || c == 0 || (int)r == N - 1 || c == N - 1)) {
                const double x = (double)c * j.h + j.min_x;
                const double y = (double)r * j.h + j.min_y;
                v = 2.0 * x * (y - 1) * (y - 2.0 * x + x * y + 2.0) * std::exp(x - y);
            }
            j.F[r * (size_t)N + c] = v;
        }
    }
}
}  // namespace

}  // extern "C"

namespace mg {
// Does k_source (exp_libm: glibc's algorithm in its FMA form, mg_kernels.hip) reproduce THIS host's libm bit for bit?
// Two grids (an odd one with an offset origin and a power of two), ~130k points, compared bitwise once per process.
// A host without FMA runs another variant of glibc's exp(): the check then fails and the host form stays in charge,
// which is that host's reference.
bool source_selfcheck()
{
    Context &c = ctx();
    if (c.source_identical >= 0) return c.source_identical == 1;
    c.source_identical = 1;
    const struct { int N; double L, mx, my; } cases[2] = {{257, 1.5, 0.25, -0.5}, {256, 1.0, 0.0, 0.0}};
    for (const auto &cs : cases) {
        const size_t n = (size_t)cs.N * cs.N;
        std::vector<double> host(n), dev(n);
        SourceJob job{cs.N, cs.L / (double)(cs.N - 1), cs.mx, cs.my, host.data()};
        source_rows(0, (size_t)cs.N, &job);
        double *d = (double *)c.pool.get(n * sizeof(double));
        if (!d) { c.source_identical = 0; break; }
        k::source_device(c.stream, cs.N, cs.L, d, cs.mx, cs.my, 0, cs.N);
        mg_download(dev.data(), d, n);
        c.pool.put(d);
        if (memcmp(host.data(), dev.data(), n * sizeof(double)) != 0) { c.source_identical = 0; break; }
    }
    return c.source_identical == 1;
}
bool source_on_device()
{
    Context &c = ctx();
    return c.source_mode == 2 || (c.source_mode == 0 && source_selfcheck());
}

void fill_source_rows(int N, double L, double min_x, double min_y, int row_lo, int row_hi, double *dev_dst)
{
    if (row_hi <= row_lo) return;
    Context &c = ctx();
    // exp_libm reproduces glibc's MAIN path (2^-54 <= |x| < 512, and the |x| < 2^-54 shortcut); past that glibc has special
    // cases (overflow, subnormal results) the device evaluates with its own exp().  In auto mode the device form is
    // therefore taken only for grids whose every argument x - y stays on the verified path: |min_x - min_y| + |L| < 512.
    const bool in_verified_range = std::fabs(min_x - min_y) + std::fabs(L) < 511.0;
    if (source_on_device() && (in_verified_range || c.source_mode == 2)) {  // no host pass, no PCIe: k_source evaluates libm's exp() algorithm on the device
        k::source_device(c.stream, N, L, dev_dst, min_x, min_y, row_lo, row_hi);
        return;
    }
    // The host evaluates (libm exp: the reference's bits) into two pinned staging buffers of at most 128 MiB
    // each and uploads chunk by chunk: while chunk i travels, chunk i+1 is computed.  The 8 GiB source of
    // N = 32768 therefore needs 256 MiB of pinned memory, not 8 GiB.
    const size_t row_bytes = (size_t)N * sizeof(double);
    size_t chunk_rows = ((size_t)128 << 20) / row_bytes;
    if (chunk_rows < 1) chunk_rows = 1;
    if (chunk_rows > (size_t)(row_hi - row_lo)) chunk_rows = (size_t)(row_hi - row_lo);
    double *stage[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    const int nbuf = chunk_rows < (size_t)(row_hi - row_lo) ? 2 : 1;
    bool ok = true;
    for (int b = 0; b < nbuf && ok; ++b)
        ok = MG_HIP(hipHostMalloc((void **)&stage[b], chunk_rows * row_bytes, hipHostMallocDefault)) && MG_HIP(hipEventCreate(&done[b]));
    int b = 0;
    for (int r0 = row_lo; r0 < row_hi && ok; r0 += (int)chunk_rows, b = (b + 1) % nbuf) {
        const int r1 = r0 + (int)chunk_rows < row_hi ? r0 + (int)chunk_rows : row_hi;
        if (r0 >= row_lo + nbuf * (int)chunk_rows) ok = MG_HIP(hipEventSynchronize(done[b]));  // the buffer's last upload
        if (!ok) break;
        // the job indexes rows globally; shift the destination so row r0 lands at stage[b][0]
        SourceJob job{N, L / (double)(N - 1), min_x, min_y, stage[b] - (size_t)r0 * N};
        struct Range { SourceJob *j; int lo; } rg{&job, r0};
        parallel_for((size_t)(r1 - r0),
                     [](size_t a, size_t e, void *arg) {
                         Range *r = (Range *)arg;
                         source_rows(a + r->lo, e + r->lo, r->j);
                     },
                     &rg, 16);
        ok = MG_HIP(hipMemcpyAsync(dev_dst + (size_t)(r0 - row_lo) * N, stage[b], (size_t)(r1 - r0) * row_bytes, hipMemcpyHostToDevice,
                                   c.stream)) &&
             MG_HIP(hipEventRecord(done[b], c.stream));
    }
    (void)hipStreamSynchronize(c.stream);
    for (int i = 0; i < 2; ++i) {
        if (stage[i]) (void)hipHostFree(stage[i]);
        if (done[i]) (void)hipEventDestroy(done[i]);
    }
}

// one fused launch on a row window (slab mode supports step <= stream_max_steps(): a second
// launch would need a ghost exchange of the intermediate field)
void slab_smooth(int N, double L, const double *U_in, double *U_out, const double *F, int step, double *raw_norm_out,
                 const SlabFusion &sf)
{
    Context &c = ctx();
    const double dx2 = spacing_sq(N, L);
    const double inv = 1.0 / dx2;
    if (step < 1 || step > k::stream_max_steps() || !k::stream_fusable(N)) {
        fail(MG_ERR_UNSUPPORTED, "row-slab mode: %d smoothing steps on N=%d (need 1..%d steps, even N)", step, N,
             k::stream_max_steps());
        return;
    }
    const ProlongTable *pt = nullptr;
    const RestrictTable *rt = nullptr;
    if (sf.coarse) {
        pt = &prolong_table(sf.Nc, N);
        if (!pt->owner_row || !pt->fusable) {
            fail(MG_ERR_UNSUPPORTED, "row-slab mode: prolongation %d -> %d is not fusable", sf.Nc, N);
            return;
        }
    }
    if (sf.Fc) {
        rt = &restrict_table(N, sf.M);
        if (!rt->lo || !rt->fusable) {
            fail(MG_ERR_UNSUPPORTED, "row-slab mode: restriction %d -> %d is not fusable", N, sf.M);
            return;
        }
    }
    const size_t n = (size_t)(sf.fine_w.own_hi - sf.fine_w.own_lo) * N;
    double bytes = (double)n * (24.0 * step + (U_in ? 0.0 : 8.0) + (sf.Fc ? 24.0 : 0.0));
    if (sf.Fc) bytes += 8.0 * n + 2.0 * n;
    if (sf.coarse) bytes += 16.0 * n + 2.0 * n;
    char name[48], pre_tag[8] = "";
    if (sf.pre) snprintf(pre_tag, sizeof pre_tag, ",pre%d", sf.pre);
    // a slab of a level the caches hold: the register-tile kernel on the row window (mg_tile_impl.h)
    const bool as_tile = !sf.pre && c.smoother == SMOOTHER_STREAM && k::tile_wanted_slab(N) && step <= k::tile_max_steps();
    snprintf(name, sizeof name, "%s<%d%s%s%s%s%s>", as_tile ? "slab_tile" : "slab_stream", step, (U_in || sf.pre) ? "" : ",zero", sf.coarse ? ",prolong" : "",
             sf.Fc ? ",res,restrict" : "", sf.no_out ? ",noU" : "", pre_tag);
    if (as_tile) {
        ProfScope ps(name, N, bytes);
        k::jacobi_tile(c.stream, N, dx2, inv, U_in, F, U_out, step, raw_norm_out, -1, sf.coarse, sf.Nc, pt, sf.Fc, sf.M, rt, sf.no_out,
                       &sf.fine_w, sf.coarse ? &sf.coarse_w : nullptr, sf.Fc ? &sf.fc_w : nullptr);
        return;
    }
    ProfScope ps(name, N, bytes);
    k::jacobi_stream(c.stream, N, dx2, inv, U_in, F, U_out, step, raw_norm_out, nullptr, -1, sf.coarse, sf.Nc, pt, sf.Fc,
                     sf.M, rt, &sf.fine_w, sf.coarse ? &sf.coarse_w : nullptr, sf.Fc ? &sf.fc_w : nullptr, sf.pre, sf.no_out);
}

void slab_smooth_f32(int N, double L, const float *U_in, float *U_out, const float *F, int step, double *raw_norm_out,
                     const SlabFusion &sf)
{
    Context &c = ctx();
    const double dx2 = spacing_sq(N, L);
    if (step < 1 || step > k::stream_max_steps() || !k::stream_fusable(N)) {
        fail(MG_ERR_UNSUPPORTED, "row-slab mode: %d smoothing steps on N=%d (need 1..%d steps, even N)", step, N,
             k::stream_max_steps());
        return;
    }
    const ProlongTable *pt = nullptr;
    const RestrictTable *rt = nullptr;
    if (sf.coarse) {
        pt = &prolong_table(sf.Nc, N);
        if (!pt->owner_row || !pt->fusable) {
            fail(MG_ERR_UNSUPPORTED, "row-slab mode: prolongation %d -> %d is not fusable", sf.Nc, N);
            return;
        }
    }
    if (sf.Fc) {
        rt = &restrict_table(N, sf.M);
        if (!rt->lo || !rt->fusable) {
            fail(MG_ERR_UNSUPPORTED, "row-slab mode: restriction %d -> %d is not fusable", N, sf.M);
            return;
        }
    }
    const size_t n = (size_t)(sf.fine_w.own_hi - sf.fine_w.own_lo) * N;
    double bytes = (double)n * (12.0 * step + (U_in ? 0.0 : 4.0) + (sf.Fc ? 12.0 : 0.0));
    if (sf.Fc) bytes += 4.0 * n + 1.0 * n;
    if (sf.coarse) bytes += 8.0 * n + 1.0 * n;
    char name[56], pre_tag[8] = "";
    if (sf.pre) snprintf(pre_tag, sizeof pre_tag, ",pre%d", sf.pre);
    const bool as_tile = !sf.pre && !sf.out_wide && c.smoother == SMOOTHER_STREAM && k::tile_wanted_slab(N) && step <= k::tile_max_steps();
    snprintf(name, sizeof name, "%s<%d%s%s%s%s%s%s>", as_tile ? "slab_tile_f32" : "slab_stream_f32", step, (U_in || sf.pre) ? "" : ",zero",
             sf.coarse ? ",prolong" : "", sf.out_wide ? ",widen" : "", sf.Fc ? ",res,restrict" : "", sf.no_out ? ",noU" : "", pre_tag);
    if (as_tile) {
        ProfScope ps(name, N, bytes);
        k::jacobi_tile_f32(c.stream, N, (float)dx2, (float)(1.0 / dx2), U_in, F, U_out, step, raw_norm_out, -1, (const float *)sf.coarse, sf.Nc, pt,
                           (float *)sf.Fc, sf.M, rt, sf.no_out, &sf.fine_w, sf.coarse ? &sf.coarse_w : nullptr, sf.Fc ? &sf.fc_w : nullptr);
        return;
    }
    ProfScope ps(name, N, bytes);
    k::jacobi_stream_f32(c.stream, N, (float)dx2, (float)(1.0 / dx2), U_in, F, U_out, step, raw_norm_out, (const float *)sf.coarse,
                         sf.Nc, pt, (float *)sf.Fc, sf.M, rt, &sf.fine_w, sf.coarse ? &sf.coarse_w : nullptr,
                         sf.Fc ? &sf.fc_w : nullptr, sf.out_wide, nullptr, -1, sf.pre, sf.no_out);
}
}  // namespace mg

extern "C" {

void mg_getSource(int N, double L, double *F, double min_x, double min_y)
{
    if (!require_ready("mg_getSource") || !grid_args_ok("mg_getSource", N)) return;
    // host libm through chunked pinned staging (the reference's bits), or k_source after mg_set_source("device")
    fill_source_rows(N, L, min_x, min_y, 0, N, F);
}

void mg_getAnalytic(int N, double L, double *U, double min_x, double min_y)
{
    if (!require_ready("mg_getAnalytic") || !grid_args_ok("mg_getAnalytic", N)) return;
    k::analytic(ctx().stream, N, L, U, min_x, min_y);
}

void mg_analyticError(int N, double L, const double *U, double min_x, double min_y, double *error_host)
{
    if (!require_ready("mg_analyticError") || !grid_args_ok("mg_analyticError", N)) return;
    Context &c = ctx();
    double *slot = c.scalars + Scalars::ANALYTIC_ERR;
    k::analytic_error(c.stream, N, L, U, min_x, min_y, slot);
    if (error_host) mg_download(error_host, slot, 1);
}

// ------------------------------------------------------------------ operators
void mg_getResidual(int N, double L, double *U, double *F, double *D)
{
    if (!require_ready("getResidual") || !grid_args_ok("getResidual", N)) return;
    ProfScope ps("residual", N, (double)N * N * 24.0);
    k::residual(ctx().stream, N, 1.0 / spacing_sq(N, L), U, F, D, +1);
}

void mg_doGridAddition(int N, double *U1, double *U2)
{
    if (!require_ready("doGridAddition") || !grid_args_ok("doGridAddition", N)) return;
    k::add(ctx().stream, (size_t)N * N, U1, U2);
}

void mg_smooth_pp(int N, double L, const double *U_in, double *U_out, double *F, int step, double *error_dev,
                  double *D_out, int d_sign)
{
    if (!require_ready("mg_smooth_pp") || !grid_args_ok("mg_smooth_pp", N)) return;
    if (U_in == U_out) {
        fail(MG_ERR_ARG, "mg_smooth_pp: U_out must differ from U_in");
        return;
    }
    smooth_pp(N, L, U_in, U_out, F, step, error_dev, D_out, d_sign < 0 ? -1 : +1);
}

// pre-smoothing + getResidual + sign flip + doRestriction of one "-1" node
// (src/MG_solver_CPU.cpp:259-287) in one pass when the streaming kernel can fuse it
void mg_smooth_restrict(int N, double L, const double *U_in, double *U_out, double *F, int step, double *error_dev,
                        int M, double *F_c)
{
    if (!require_ready("mg_smooth_restrict") || !grid_args_ok("mg_smooth_restrict", N) ||
        !grid_args_ok("mg_smooth_restrict", M))
        return;
    if (U_in == U_out) {
        fail(MG_ERR_ARG, "mg_smooth_restrict: U_out must differ from U_in");
        return;
    }
    Context &c = ctx();
    const RestrictTable &rt = restrict_table(N, M);
    if (!rt.lo) return;
    if (step > 0 && c.smoother != SMOOTHER_SIMPLE && k::stream_fusable(N) && rt.fusable) {
        Fusion fu;
        fu.Fc = F_c;
        fu.M = M;
        fu.rt = &rt;
        smooth_pp(N, L, U_in, U_out, F, step, error_dev, nullptr, -1, fu);
        return;
    }
    // operator by operator, D in pool scratch
    const size_t n = (size_t)N * N;
    double *D = (double *)scratch_pool().get(n * sizeof(double));
    if (!D) return;
    smooth_pp(N, L, U_in, U_out, F, step, error_dev, D, -1);
    {
        ProfScope ps("restrict", N, 8.0 * N * N + 8.0 * M * M);
        k::restrict_gather(c.stream, N, D, M, F_c, rt, +1);
    }
    scratch_pool().put(D);
}

// doProlongation + doGridAddition + post-smoothing of one "1" node
// (src/MG_solver_CPU.cpp:353-416): U_out = smooth^step(U_in + P(U_c))
void mg_prolong_smooth(int Nc, const double *U_c, int N, double L, const double *U_in, double *U_out, double *F,
                       int step, double *error_dev)
{
    if (!require_ready("mg_prolong_smooth") || !grid_args_ok("mg_prolong_smooth", N) ||
        !grid_args_ok("mg_prolong_smooth", Nc))
        return;
    if (U_in == U_out) {
        fail(MG_ERR_ARG, "mg_prolong_smooth: U_out must differ from U_in");
        return;
    }
    Context &c = ctx();
    const ProlongTable &pt = prolong_table(Nc, N);
    if (!pt.owner_row) return;
    if (step > 0 && c.smoother != SMOOTHER_SIMPLE && k::stream_fusable(N) && pt.fusable) {
        Fusion fu;
        fu.coarse = U_c;
        fu.Nc = Nc;
        fu.pt = &pt;
        smooth_pp(N, L, U_in, U_out, F, step, error_dev, nullptr, +1, fu);
        return;
    }
    {
        ProfScope ps("prolong_add", N, 8.0 * Nc * Nc + 16.0 * N * N);
        k::prolong(c.stream, Nc, U_c, N, U_in, U_out, pt);
    }
    if (step > 0) {
        // U_out now holds U + P; sweep it through U_in (clobbered) and back
        const size_t n = (size_t)N * N;
        double *tmp = (double *)scratch_pool().get(n * sizeof(double));
        if (!tmp) return;
        smooth_pp(N, L, U_out, tmp, F, step, error_dev, nullptr, +1);
        MG_HIP(hipMemcpyAsync(U_out, tmp, n * sizeof(double), hipMemcpyDeviceToDevice, c.stream));
        scratch_pool().put(tmp);
    }
}

// ------------------------------------------------------------------ mixed precision (fp32 fields)
// The fp32 forms of the two fused nodes: every array and every arithmetic operation in fp32,
// norms in fp64.  They exist for the streaming smoother on even N with fusable tables -- the
// shapes a halving hierarchy produces; anything else is refused (no silent fp64 fallback).
void mg_smooth_restrict_f32(int N, double L, const float *U_in, float *U_out, float *F, int step, double *error_dev,
                            int M, float *F_c)
{
    if (!require_ready("mg_smooth_restrict_f32") || !grid_args_ok("mg_smooth_restrict_f32", N) ||
        !grid_args_ok("mg_smooth_restrict_f32", M))
        return;
    Context &c = ctx();
    const RestrictTable &rt = restrict_table(N, M);
    if (!rt.lo) return;
    if (U_in != nullptr || step < 1 || step > k::stream_max_steps()) {
        fail(MG_ERR_UNSUPPORTED, "mg_smooth_restrict_f32: needs a zero start and 1..%d steps (N=%d M=%d step=%d)",
             k::stream_max_steps(), N, M, step);
        return;
    }
    const double dx2 = spacing_sq(N, L);
    const size_t n = (size_t)N * N;
    if (k::stream_fusable(N) && rt.fusable) {
        if (c.smoother == SMOOTHER_STREAM && k::tile_wanted(N) && step <= k::tile_max_steps()) {  // small levels: mg_tile_impl.h
            ProfScope ps("jacobi_tile_f32<zero,res,restrict>", N, (double)n * (12.0 * step + 4.0 + 12.0 + 4.0) + 4.0 * M * M);
            k::jacobi_tile_f32(c.stream, N, (float)dx2, (float)(1.0 / dx2), nullptr, F, U_out, step, error_dev, -1, nullptr, 0, nullptr, F_c, M,
                               &rt, false);
            return;
        }
        ProfScope ps("jacobi_stream_f32<zero,res,restrict>", N, (double)n * (12.0 * step + 4.0 + 12.0 + 4.0) + 4.0 * M * M);
        k::jacobi_stream_f32(c.stream, N, (float)dx2, (float)(1.0 / dx2), nullptr, F, U_out, step, error_dev, nullptr, 0, nullptr, F_c,
                             M, &rt);
        return;
    }
    // odd or non-nested sizes: operator by operator (sweeps + signed residual in one launch, then the gather)
    float *D = (float *)scratch_pool().get(n * sizeof(float));
    if (!D) return;
    {
        ProfScope ps("jacobi_stream_f32<zero,res>", N, (double)n * (12.0 * step + 4.0 + 12.0));
        k::jacobi_stream_f32(c.stream, N, (float)dx2, (float)(1.0 / dx2), nullptr, F, U_out, step, error_dev, nullptr, 0, nullptr,
                             nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, D, -1);
    }
    {
        ProfScope ps("restrict_f32", N, 4.0 * N * N + 4.0 * M * M);
        k::restrict_gather_f32(c.stream, N, D, M, F_c, rt, +1);
    }
    scratch_pool().put(D);
}

}  // extern "C"

namespace mg {
namespace {
void prolong_smooth_f32_impl(int Nc, const float *U_c, int N, double L, const float *U_in, float *U_out, double *U_out_wide,
                             const float *F, int step, double *error_dev, int pre = 0)
{
    if (!require_ready("mg_prolong_smooth_f32") || !grid_args_ok("mg_prolong_smooth_f32", N) ||
        !grid_args_ok("mg_prolong_smooth_f32", Nc))
        return;
    Context &c = ctx();
    const ProlongTable &pt = prolong_table(Nc, N);
    if (!pt.owner_row) return;
    if ((!U_out_wide && U_in == U_out) || step < 1 || step > k::stream_max_steps()) {
        fail(MG_ERR_UNSUPPORTED, "mg_prolong_smooth_f32: needs 1..%d steps and U_out != U_in (Nc=%d N=%d step=%d)",
             k::stream_max_steps(), Nc, N, step);
        return;
    }
    const double dx2 = spacing_sq(N, L);
    const size_t n = (size_t)N * N;
    if (pre && (!k::stream_fusable(N) || !pt.fusable)) {
        fail(MG_ERR_UNSUPPORTED, "prolong_smooth_f32_recompute: needs the fused form (Nc=%d N=%d)", Nc, N);
        return;
    }
    if (!k::stream_fusable(N) || !pt.fusable) {
        // odd or non-nested sizes: prolongation + addition as a gather, then the sweeps
        if (U_out_wide) {
            fail(MG_ERR_UNSUPPORTED, "mg_prolong_smooth_f32: the widening store needs the fused form (Nc=%d N=%d)", Nc, N);
            return;
        }
        float *tmp = (float *)scratch_pool().get(n * sizeof(float));
        if (!tmp) return;
        {
            ProfScope ps("prolong_add_f32", N, 4.0 * Nc * Nc + 8.0 * N * N);
            k::prolong_add_f32(c.stream, Nc, U_c, N, U_in, tmp, pt);
        }
        {
            ProfScope ps("jacobi_stream_f32", N, (double)n * 12.0 * step);
            k::jacobi_stream_f32(c.stream, N, (float)dx2, (float)(1.0 / dx2), tmp, F, U_out, step, error_dev, nullptr, 0, nullptr, nullptr,
                                 0, nullptr);
        }
        scratch_pool().put(tmp);
        return;
    }
    if (!pre && !U_out_wide && c.smoother == SMOOTHER_STREAM && k::tile_wanted(N) && step <= k::tile_max_steps()) {  // small levels: mg_tile_impl.h
        ProfScope ps("jacobi_tile_f32<prolong>", N, (double)n * (12.0 * step + 8.0) + 4.0 * Nc * Nc);
        k::jacobi_tile_f32(c.stream, N, (float)dx2, (float)(1.0 / dx2), U_in, F, U_out, step, error_dev, +1, U_c, Nc, &pt, nullptr, 0, nullptr,
                           false);
        return;
    }
    char name[48];
    snprintf(name, sizeof name, pre ? "jacobi_stream_f32<prolong%s,pre%d>" : "jacobi_stream_f32<prolong%s>", U_out_wide ? ",widen" : "", pre);
    ProfScope ps(name,
                 N, (double)n * (12.0 * (step + pre) + 8.0 + (U_out_wide ? 12.0 : 0.0)) + 4.0 * Nc * Nc);
    k::jacobi_stream_f32(c.stream, N, (float)dx2, (float)(1.0 / dx2), pre ? nullptr : U_in, F, U_out, step, error_dev, U_c, Nc, &pt, nullptr, 0,
                         nullptr, nullptr, nullptr, nullptr, U_out_wide, nullptr, -1, pre, false);
}
}  // namespace

void smooth_restrict_f32_no_out(int N, double L, float *U_unused, float *F, int step, double *error_dev, int M, float *F_c)
{
    Context &c = ctx();
    const RestrictTable &rt = restrict_table(N, M);
    if (!rt.lo || !rt.fusable || !k::stream_fusable(N) || step < 1 || step > k::stream_max_steps()) {
        fail(MG_ERR_UNSUPPORTED, "smooth_restrict_f32_no_out: N=%d M=%d step=%d is not a fused `-1` node", N, M, step);
        return;
    }
    const double dx2 = spacing_sq(N, L);
    const size_t n = (size_t)N * N;
    ProfScope ps("jacobi_stream_f32<restrict,noU>", N, (double)n * (12.0 * step + 4.0 + 12.0 + 4.0) + 4.0 * M * M);
    k::jacobi_stream_f32(c.stream, N, (float)dx2, (float)(1.0 / dx2), nullptr, F, U_unused, step, error_dev, nullptr, 0, nullptr, F_c, M, &rt,
                         nullptr, nullptr, nullptr, nullptr, nullptr, -1, 0, true);
}

void prolong_smooth_f32_recompute(int Nc, const float *U_c, int N, double L, float *U_out, double *U_out_wide, const float *F, int pre,
                                  int step, double *error_dev)
{
    if (!recompute_available(Nc, N, pre, step)) {
        fail(MG_ERR_UNSUPPORTED, "prolong_smooth_f32_recompute: Nc=%d N=%d pre=%d step=%d", Nc, N, pre, step);
        return;
    }
    prolong_smooth_f32_impl(Nc, U_c, N, L, nullptr, U_out, U_out_wide, F, step, error_dev, pre);
}

void prolong_smooth_f32_wide(int Nc, const float *U_c, int N, double L, const float *U_in, double *U_out_wide, const float *F,
                             int step, double *error_dev)
{
    prolong_smooth_f32_impl(Nc, U_c, N, L, U_in, nullptr, U_out_wide, F, step, error_dev);
}
}  // namespace mg

extern "C" {

void mg_prolong_smooth_f32(int Nc, const float *U_c, int N, double L, const float *U_in, float *U_out, float *F, int step,
                           double *error_dev)
{
    prolong_smooth_f32_impl(Nc, U_c, N, L, U_in, U_out, nullptr, F, step, error_dev);
}

float *mg_alloc_f32(size_t n)
{
    if (!require_ready("mg_alloc_f32")) return nullptr;
    return (float *)ctx().pool.get(n * sizeof(float));
}
void mg_free_f32(float *p)
{
    if (!p || !require_ready("mg_free_f32")) return;
    ctx().pool.put(p);
}
void mg_to_f32(float *dst, const double *src, size_t n)
{
    if (!require_ready("mg_to_f32")) return;
    k::convert_to_f32(ctx().stream, dst, src, n);
}
void mg_to_f64(double *dst, const float *src, size_t n)
{
    if (!require_ready("mg_to_f64")) return;
    k::convert_to_f64(ctx().stream, dst, src, n);
}
void mg_upload_f32(float *dev, const float *host, size_t n)
{
    if (!require_ready("mg_upload_f32")) return;
    hipStream_t s = ctx().stream;
    if (MG_HIP(hipMemcpyAsync(dev, host, n * sizeof(float), hipMemcpyHostToDevice, s))) MG_HIP(hipStreamSynchronize(s));
}
void mg_download_f32(float *host, const float *dev, size_t n)
{
    if (!require_ready("mg_download_f32")) return;
    hipStream_t s = ctx().stream;
    if (MG_HIP(hipMemcpyAsync(host, dev, n * sizeof(float), hipMemcpyDeviceToHost, s))) MG_HIP(hipStreamSynchronize(s));
}

void mg_doSmoothing(int N, double L, double *U, double *F, int step, double *error)
{
    if (!require_ready("doSmoothing") || !grid_args_ok("doSmoothing", N)) return;
    Context &c = ctx();
    const size_t n = (size_t)N * N;
    double *slot = c.scalars + Scalars::SMOOTH_ERR;
    const bool deferred = c.defer_norms;  // this entry point hands the value back: finish now
    c.defer_norms = false;
    if (step > 0) {
        // in-place semantics of the reference on top of the out-of-place kernels:
        // sweep into pool scratch, then hand the result back into U
        double *tmp = (double *)scratch_pool().get(n * sizeof(double));
        if (!tmp) {
            c.defer_norms = deferred;
            return;
        }
        smooth_pp(N, L, U, tmp, F, step, error ? slot : nullptr, nullptr, +1);
        MG_HIP(hipMemcpyAsync(U, tmp, n * sizeof(double), hipMemcpyDeviceToDevice, c.stream));
        scratch_pool().put(tmp);  // stream-ordered reuse: later users enqueue behind the copy
    } else if (error) {
        k::smoothing_error(c.stream, N, 1.0 / spacing_sq(N, L), U, F, slot);
    }
    c.defer_norms = deferred;
    if (error) mg_download(error, slot, 1);
}

void mg_doExactSolver(int N, double L, double *U, double *F, double target_error, int option)
{
    if (!require_ready("doExactSolver") || !grid_args_ok("doExactSolver", N)) return;
    if (option == 0) {  // src/MG_solver_GPU.cu:1286-1289
        fail(MG_ERR_UNSUPPORTED, "doExactSolver: the Inverse Matrix solver (option 0) is not available on the GPU path; use option 1");
        return;
    }
    if (option != 1) return;  // the reference silently does nothing for other options (:630-637)
    const double h2 = spacing_sq(N, L);
    k::gauss_seidel(ctx().stream, N, h2, 1.0 / h2, U, F, target_error, ctx().gs_state);
}

int mg_lastExactSolverIterations(void)
{
    if (!require_ready("mg_lastExactSolverIterations")) return -1;
    Context &c = ctx();
    if (!MG_HIP(hipMemcpyAsync(c.host_ints, c.gs_state, 2 * sizeof(int), hipMemcpyDeviceToHost, c.stream))) return -1;
    if (!MG_HIP(hipStreamSynchronize(c.stream))) return -1;
    return c.host_ints[1];
}

void mg_restrict_signed(int N, const double *U_f, int M, double *U_c, int sign)
{
    if (!require_ready("doRestriction") || !grid_args_ok("doRestriction", N) || !grid_args_ok("doRestriction", M)) return;
    const RestrictTable &t = restrict_table(N, M);
    if (!t.lo) return;
    ProfScope ps("restrict", N, 8.0 * N * N + 8.0 * M * M);
    k::restrict_gather(ctx().stream, N, U_f, M, U_c, t, sign < 0 ? -1 : +1);
}

void mg_doRestriction(int N, double *U_f, int M, double *U_c) { mg_restrict_signed(N, U_f, M, U_c, +1); }

void mg_doProlongation(int N, double *U_c, int M, double *U_f)
{
    if (!require_ready("doProlongation") || !grid_args_ok("doProlongation", N) || !grid_args_ok("doProlongation", M)) return;
    const ProlongTable &t = prolong_table(N, M);
    if (!t.owner_row) return;
    ProfScope ps("prolong", M, 8.0 * N * N + 8.0 * M * M);
    k::prolong(ctx().stream, N, U_c, M, nullptr, U_f, t);
}

void mg_prolongAdd(int N, const double *U_c, int M, const double *U_f_in, double *U_f_out)
{
    if (!require_ready("mg_prolongAdd") || !grid_args_ok("mg_prolongAdd", N) || !grid_args_ok("mg_prolongAdd", M)) return;
    const ProlongTable &t = prolong_table(N, M);
    if (!t.owner_row) return;
    ProfScope ps("prolong_add", M, 8.0 * N * N + 16.0 * M * M);
    k::prolong(ctx().stream, N, U_c, M, U_f_in, U_f_out, t);
}

// ------------------------------------------------------------------ live timing
void mg_profile_begin(int min_N)
{
    if (!require_ready("mg_profile_begin")) return;
    Context &c = ctx();
    (void)hipStreamSynchronize(c.stream);
    for (auto &r : c.prof) {
        c.event_pool.push_back(r.e0);
        c.event_pool.push_back(r.e1);
    }
    c.prof.clear();
    c.profile_min_N = min_N;
    c.profile_window = 0;
    c.profiling = true;
}

// time the launches of every `every`-th cycle window only (the first, the (every+1)-th, ...): an event pair costs its
// launch a few microseconds, which a benchmark does not want on every one of its timed windows.  1 = every window.
void mg_profile_sample(int every) { ctx().profile_every = every < 1 ? 1 : every; }

int mg_profile_end(mg_profile_entry *out, int cap)
{
    if (!require_ready("mg_profile_end")) return 0;
    Context &c = ctx();
    c.profiling = false;
    (void)hipStreamSynchronize(c.stream);
    int count = 0;
    for (auto &r : c.prof) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) {
            (void)hipGetLastError();
            ms = 0.f;
        }
        int j = 0;
        for (; j < count; ++j)
            if (out[j].N == r.N && r.name == out[j].name) break;
        if (j == count) {
            if (count >= cap) continue;
            memset(&out[j], 0, sizeof out[j]);
            snprintf(out[j].name, sizeof out[j].name, "%s", r.name.c_str());
            out[j].N = r.N;
            out[j].algo_bytes = r.bytes;
            ++count;
        }
        out[j].launches += 1;
        out[j].total_ms += ms;
        c.event_pool.push_back(r.e0);
        c.event_pool.push_back(r.e1);
    }
    c.prof.clear();
    return count;
}

int mg_recompute_pair_available(int pre, int post) { return k::stream_recompute_supported(pre, post) ? 1 : 0; }

// ------------------------------------------------------------------ tables
void mg_restriction_table(int N, int M, int *lo, double *w) { build_restriction_table(N, M, lo, w); }
void mg_prolongation_table(int N, int M, int axis, int *owner, double *w_hi, double *w_lo)
{
    build_prolongation_table(N, M, axis, owner, w_hi, w_lo);
}

// ------------------------------------------------------------------ synthetic data
void mg_fill_uniform(double *dst, size_t n, uint64_t seed)
{
    if (!require_ready("mg_fill_uniform")) return;
    k::fill_uniform(ctx().stream, dst, n, seed);
}

void mg_checksum(const double *src, size_t n, uint64_t out[2])
{
    if (!require_ready("mg_checksum")) return;
    Context &c = ctx();
    uint64_t *slot = (uint64_t *)(c.scalars + Scalars::CHECKSUM);
    k::checksum(c.stream, src, n, slot);
    if (MG_HIP(hipMemcpyAsync(c.host_scalars, slot, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, c.stream)) &&
        MG_HIP(hipStreamSynchronize(c.stream))) {
        memcpy(out, c.host_scalars, 2 * sizeof(uint64_t));
    }
}

}  // extern "C"
