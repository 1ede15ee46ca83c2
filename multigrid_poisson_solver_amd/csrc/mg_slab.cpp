// mg_slab.cpp -- the cycle-file driver on a 1-D row-slab decomposition.
//
// New work with no counterpart in the reference (single device).  The
// fine levels of the hierarchy are cut into contiguous row slabs, one per rank; a slab is one
// contiguous block of the row-major array and a ghost row is one contiguous message.  Each
// slab carries halo rows on either side (per level as many as the schedule below needs).
//
// Because the smoother is temporally blocked, one set of halo rows feeds a whole fused node (S sweeps +
// residual + restriction, or prolongation + S sweeps), and because xGMI exchanges are LATENCY-bound
// (a ghost message is a few hundred KiB; an RCCL group costs tens of microseconds whatever it carries)
// the schedule is communication-avoiding:
//   * way DOWN: a `-1` launch updates not only the rows its rank owns but as many rows beyond them as the
//     NEXT level's launches will read of the restricted residual (halo of the next level's F): those rows
//     are recomputed redundantly instead of exchanged.  The finest F comes from getSource, which every rank
//     evaluates on its whole window, so the descent needs NO exchange while the extra rows stay below a
//     share of the slab (MG_SLAB_CA_PCT, default 10 %: beyond that the next level's F halo is exchanged
//     instead, which resets the growth);
//   * way UP: a `1` node also updates the few halo rows of its level that the next finer level's
//     prolongation will read (at most 7 rows per side, the fixed point of e -> (e + S + 2)/2 + 1), from
//     inputs that are already there.  The halo of U that a `1` node reads is either part of what the
//     descent computed redundantly or travels in ONE exchange issued right after the level's `-1` launch on
//     a second stream: it is needed only when the cycle comes back up, so it overlaps with the whole
//     coarser part of the cycle.
// Redundant rows are bit-identical to the neighbour's owned rows (same inputs, same arithmetic).
// Levels at or below collapse_N are collapsed: after the last
// distributed restriction every rank hands its rows of that level's F to the others, and
// EVERY rank -- rank 0 included -- runs that part of the cycle file on the whole coarse grid
// with the single-GPU operators.  Replicating the collapsed levels instead of parking them on
// rank 0 costs nothing (the other ranks would idle) and removes the broadcast of the coarse
// correction from the critical path; all ranks hold bit-identical copies.  That all-gather is the
// one communication step on the critical path of a V-cycle.
//
// Ranks may all live in THIS process ("virtual ranks": exchanges are device-to-device
// copies) -- that is how the decomposition is tested bit-for-bit on a one-GPU box -- or one
// per process over RCCL (mg_comm.cpp), which is how bench.py --gpus N runs.
//
// The coarse partition is induced by the fine one: a rank owns the coarse rows whose
// lower-left restriction sample (doRestriction's iy_f, src/MG_solver_CPU.cpp:662) lies in
// its fine rows, so the fused restriction never writes a remote row.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "mg_internal.h"

namespace mg {
namespace {

// A level stays distributed while every slab keeps at least 2 * MIN_HALF rows (S = 4 sweeps per node: up to
// 8 redundant rows on the way up + the S+2 rows a launch reads beyond them = 14 per side); how many halo
// rows a level's arrays really carry is decided by the schedule (LevelPlan::halo).
constexpr int MIN_HALF = 14;

struct Span {  // rows [lo, hi)
    int lo = 0, hi = 0;
};
inline Span clip(Span s, int N) { return Span{std::max(0, s.lo), std::min(N, s.hi)}; }
inline Span unite(Span a, Span b) { return Span{std::min(a.lo, b.lo), std::max(a.hi, b.hi)}; }
inline Span grow(Span s, int h, int N) { return clip(Span{s.lo - h, s.hi + h}, N); }
inline bool inside(Span a, Span b) { return a.lo >= b.lo && a.hi <= b.hi; }  // a within b
inline int beyond(Span s, Span own) { return std::max(own.lo - s.lo, s.hi - own.hi); }

// what one distributed level does on every rank (per GLOBAL rank)
struct LevelPlan {
    std::vector<Span> own;   // rows the rank owns
    std::vector<Span> dext;  // rows its `-1` launch updates: own + what the next level's F halo needs (recomputed, not exchanged)
    std::vector<Span> ext;   // rows its `1` launch updates: own + what the next finer level's prolongation reads
    std::vector<Span> fwr;   // rows of THIS level's F that the finer level's `-1` launch of the rank writes
    int halo = 0;            // halo rows per side the level's arrays carry
    int needF = 0;           // rows of F beyond the owned rows that the level's launches read
    int xF = 0;              // > 0: that many rows of this level's F are exchanged after the finer level's `-1` launch
    int xU = 0;              // > 0: that many rows of this level's U are exchanged after its own `-1` launch
    int pre = 0;             // > 0: the level's `-1` launch does not store U, its `1` launch recomputes it (pre sweeps from zero on F)
};

struct Partition {
    std::vector<int> lo, hi;  // rows [lo[r], hi[r]) owned by global rank r
};

Partition split_rows(int N, int R)
{
    Partition p;
    const int base = N / R, rem = N % R;
    int at = 0;
    for (int r = 0; r < R; ++r) {
        const int n = base + (r < rem ? 1 : 0);
        p.lo.push_back(at);
        at += n;
        p.hi.push_back(at);
    }
    return p;
}

// coarse rows owned by the rank whose fine rows hold their lower-left sample lo[rc]; the rim
// rows 0 and M-1 (never sampled, always zero) go to the first and last rank
Partition induced_partition(const Partition &fine, int N, int M)
{
    std::vector<int> lo((size_t)M);
    std::vector<double> w((size_t)M);
    build_restriction_table(N, M, lo.data(), w.data());
    const int R = (int)fine.lo.size();
    Partition c;
    int rc = 1;
    for (int r = 0; r < R; ++r) {
        c.lo.push_back(rc);
        while (rc <= M - 2 && lo[(size_t)rc] < fine.hi[(size_t)r]) ++rc;
        c.hi.push_back(rc);
    }
    c.lo[0] = 0;
    c.hi[(size_t)R - 1] = M;
    return c;
}

int min_rows(const Partition &p)
{
    int m = 1 << 30;
    for (size_t r = 0; r < p.lo.size(); ++r) m = std::min(m, p.hi[r] - p.lo[r]);
    return m;
}

struct Local {  // one local rank's arrays of one level
    double *U = nullptr, *F = nullptr, *D = nullptr;
};

struct Level {
    int N = 0;
    int hier = 0;             // index into the plan's hierarchy (sizes / parts / lp)
    int halo = 0;             // halo rows per side of a distributed level's windows
    bool collapsed = false;   // whole grid replicated on every rank
    Partition part;           // distributed levels
    std::vector<Local> loc;   // per local rank (collapsed: full N x N arrays, distributed: windows)
};

}  // namespace
}  // namespace mg

using namespace mg;

struct mg_slab_plan {
    std::string path;
    double L = 1.0, min_x = 0.0, min_y = 0.0;
    int con_step = 0, con_N = 0, N_max = 0, N_min = 0;
    std::vector<int> sizes;
    std::vector<double> tokens;
    int nranks = 1;
    std::vector<int> local;   // global ranks handled by this process
    bool real = false;        // one rank per process over RCCL
    int collapse_N = 512;
    Pool pool;
    std::vector<Level> levels;
    std::vector<Partition> parts;       // per hierarchy index (sizes[i]), distributed ones only
    std::vector<bool> level_collapsed;  // per hierarchy index
    std::vector<mg_node_record> records;
    std::vector<int> rec_final;         // 1: the root's slot already holds the finished value
    double *raw_dev = nullptr;          // [max_rec][n_local] raw sums (+1 row for the analytic error)
    double *all_dev = nullptr;          // real mode: allgather target [nranks][max_rec+1]
    size_t max_rec = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int status = 0;
    int want_error = 1;   // evaluate the analytic error after the window (outside the timing)
    bool mixed = false;   // MG_CYCLE_MIXED: fp32 fields (the double* members then only carry addresses)
    size_t elem = sizeof(double);
    std::vector<double *> U64;   // mixed: per local rank, the finest window in fp64 (result / error / refinement iterate)
    std::vector<double *> F64;   // mixed: per local rank, the fp64 source rows of the finest window
    int refinements = 1;         // mixed: fp32 cycles per window, joined by an fp64 residual and correction
    bool U64_current = false;    // U64 holds the result of the last window (refinement keeps it up to date)
    int refine_it = 0;           // mixed, refinement: which cycle of the window is running
    bool top_widened = false;    // ... and its last node stored the owned rows of the result in U64 itself
    bool F32_stale = false;      // the finest fp32 F holds a residual, not the rounded source
    std::vector<double *> F32_res;  // mixed, refinement: per local rank an fp32 window of its own for the residual source of the
                                    // correction cycles (the rounded source of the first cycle then never has to be made again)
    std::vector<double *> F32_src;  // ... and the finest level's own F arrays while a correction cycle points at F32_res
    double *refine_raw = nullptr;  // [refinements-1][n_local] raw residual norms of the intermediate iterates
    double *refine_all = nullptr;  // real mode: all-gather target
    std::vector<double> refine_err;
    std::vector<LevelPlan> lp;   // per hierarchy index: the schedule of the distributed levels (slab_schedule)
    bool poison = false;         // MG_SLAB_POISON: fresh level arrays are filled with NaN (tests)
    // exchanges run on a second stream: one event pair per hierarchy index (launch done -> exchange may start,
    // exchange done -> the consumer may start); pending[l]: level l's U halo is still on its way
    hipStream_t comm = nullptr;
    std::vector<hipEvent_t> ev_ready, ev_done;
    std::vector<char> pending;
};

namespace {

RowWindow window_of(const Level &lv, int r)
{
    RowWindow w;
    w.own_lo = lv.part.lo[(size_t)r];
    w.own_hi = lv.part.hi[(size_t)r];
    w.base = w.own_lo - lv.halo;
    w.rows = (w.own_hi - w.own_lo) + 2 * lv.halo;
    return w;
}

double *row_ptr(double *a, const RowWindow &w, int N, int y) { return a + (size_t)(y - w.base) * N; }
// the same for a plan's arrays, whose element size depends on the mode
void *row_at(const mg_slab_plan *p, double *a, const RowWindow &w, int N, int y)
{
    return (char *)a + (size_t)(y - w.base) * N * p->elem;
}

void alloc_level(mg_slab_plan *p, Level &lv)
{
    lv.loc.resize(p->local.size());
    if (lv.collapsed) {
        const size_t bytes = (size_t)lv.N * lv.N * p->elem;
        for (Local &l : lv.loc) {
            l.U = (double *)p->pool.get(bytes);
            l.F = (double *)p->pool.get(bytes);
            l.D = (double *)p->pool.get(bytes);
        }
        return;
    }
    for (size_t i = 0; i < p->local.size(); ++i) {
        const RowWindow w = window_of(lv, p->local[i]);
        const size_t bytes = (size_t)w.rows * lv.N * p->elem;
        lv.loc[i].U = (double *)p->pool.get(bytes);
        lv.loc[i].F = (double *)p->pool.get(bytes);
        lv.loc[i].D = (double *)p->pool.get(bytes);
        if (p->poison) {  // all-ones bytes are a NaN in fp64 and in fp32: a row nobody wrote shows up in the result
            for (double *a : {lv.loc[i].U, lv.loc[i].F, lv.loc[i].D})
                if (a) (void)hipMemsetAsync(a, 0xFF, bytes, ctx().stream);
        }
    }
}

void free_level(mg_slab_plan *p, Level &lv)
{
    for (Local &l : lv.loc) {
        if (l.U) p->pool.put(l.U);
        if (l.F) p->pool.put(l.F);
        if (l.D) p->pool.put(l.D);
    }
    lv.loc.clear();
}

// ghost rows of arrays of distributed levels: every rank sends its top `depth` owned rows up and
// its bottom `depth` owned rows down, and receives the neighbours' into its halo.  All arrays of
// one call travel in ONE RCCL group (one launch, both neighbours, both directions).
enum Which { ARR_U, ARR_F };
struct GhostItem {
    Level *lv;
    Which which;
    int depth;  // rows per side
    // other arrays with the level's window geometry (the fp64 iterate of the refinement): one per local rank
    const std::vector<double *> *raw = nullptr;
    size_t raw_elem = 0;
};
void share_rows(mg_slab_plan *p, Level &coarse, const Partition &cpart, bool in_open_group, hipStream_t s);

// Exchanges run on the plan's second stream: it first waits for what the engine's stream has enqueued so far
// (the launch that produced the rows), and whoever consumes the received rows waits for ev_done[slot] -- at
// once for rows on the critical path (a next level's F halo, the collapse all-gather), not before the cycle
// comes back up through the level for a U halo (exchange_join), which hides that exchange behind the whole
// coarser part of the cycle.
// `share`/`share_part`: the collapse all-gather of the next level's F rides in the same group
void exchange_ghosts(mg_slab_plan *p, size_t slot, const std::vector<GhostItem> &items, Level *share = nullptr,
                     const Partition *share_part = nullptr)
{
    Context &c = ctx();
    const int R = p->nranks;
    if (R == 1) return;
    if (items.empty() && !share) return;
    hipStream_t s = p->comm;
    (void)hipEventRecord(p->ev_ready[slot], c.stream);
    (void)hipStreamWaitEvent(s, p->ev_ready[slot], 0);
    {
        // live timing like every kernel launch (bench.py --gpus N reports it per level): event pair around the group
        size_t bytes = 0;
        for (const GhostItem &it : items) bytes += (size_t)2 * it.depth * it.lv->N * (it.raw ? it.raw_elem : p->elem);
        ProfScope ps(share ? "ghost_exchange+collapse_allgather" : "ghost_exchange", items.empty() ? share->N : items[0].lv->N, (double)bytes, s);
        comm_set_stream(s);
        if (p->real) comm_group_begin();
        for (const GhostItem &it : items) {
            Level &lv = *it.lv;
            const int N = lv.N;
            const size_t elem = it.raw ? it.raw_elem : p->elem;
            const int G = it.depth;
            const size_t cnt = (size_t)G * N * elem;  // bytes
            auto arr = [&](size_t i) { return it.raw ? (*it.raw)[i] : (it.which == ARR_U ? lv.loc[i].U : lv.loc[i].F); };
            auto rows_from = [&](double *a, const RowWindow &w, int y) {  // row y of an array with this item's element size
                return (void *)((char *)a + (size_t)(y - w.base) * N * elem);
            };
            if (!p->real) {
                for (int r = 0; r + 1 < R; ++r) {  // pair (r, r+1), both local
                    const RowWindow a = window_of(lv, r), b = window_of(lv, r + 1);
                    double *A = arr((size_t)r), *B = arr((size_t)r + 1);
                    // a's top owned rows -> b's lower halo; b's bottom owned rows -> a's upper halo
                    (void)hipMemcpyAsync(rows_from(B, b, b.own_lo - G), rows_from(A, a, a.own_hi - G), cnt, hipMemcpyDeviceToDevice, s);
                    (void)hipMemcpyAsync(rows_from(A, a, a.own_hi), rows_from(B, b, b.own_lo), cnt, hipMemcpyDeviceToDevice, s);
                }
                continue;
            }
            const int r = p->local[0];
            const RowWindow w = window_of(lv, r);
            double *A = arr(0);
            if (r + 1 < R) {
                comm_send(rows_from(A, w, w.own_hi - G), cnt, r + 1);
                comm_recv(rows_from(A, w, w.own_hi), cnt, r + 1);
            }
            if (r > 0) {
                comm_send(rows_from(A, w, w.own_lo), cnt, r - 1);
                comm_recv(rows_from(A, w, w.own_lo - G), cnt, r - 1);
            }
        }
        if (share) share_rows(p, *share, *share_part, true, s);
        if (p->real) comm_group_end();
        comm_set_stream(nullptr);
    }
    (void)hipEventRecord(p->ev_done[slot], s);
    p->pending[slot] = 1;
}

// the engine's stream waits for the exchange of `slot` (no-op when none is outstanding)
void exchange_join(mg_slab_plan *p, size_t slot)
{
    if (slot >= p->pending.size() || !p->pending[slot]) return;
    (void)hipStreamWaitEvent(ctx().stream, p->ev_done[slot], 0);
    p->pending[slot] = 0;
}

// collapse boundary: every rank wrote its rows [cpart.lo, cpart.hi) of the coarse F into its own
// full M x M array; afterwards every rank holds all rows (an all-gather with per-rank row
// counts, issued as one group of point-to-point transfers)
void share_rows(mg_slab_plan *p, Level &coarse, const Partition &cpart, bool in_open_group, hipStream_t s)
{
    const int M = coarse.N, R = p->nranks;
    if (R == 1) return;
    auto off = [&](int r) { return (size_t)cpart.lo[(size_t)r] * M * p->elem; };                            // bytes
    auto cnt = [&](int r) { return (size_t)(cpart.hi[(size_t)r] - cpart.lo[(size_t)r]) * M * p->elem; };  // bytes
    if (!p->real) {
        for (int src = 0; src < R; ++src)
            for (int dst = 0; dst < R; ++dst)
                if (src != dst)
                    (void)hipMemcpyAsync((char *)coarse.loc[(size_t)dst].F + off(src), (char *)coarse.loc[(size_t)src].F + off(src),
                                         cnt(src), hipMemcpyDeviceToDevice, s);
        return;
    }
    const int me = p->local[0];
    char *F = (char *)coarse.loc[0].F;
    if (!in_open_group) comm_group_begin();
    for (int r = 0; r < R; ++r) {
        if (r == me) continue;
        comm_send(F + off(me), cnt(me), r);
        comm_recv(F + off(r), cnt(r), r);
    }
    if (!in_open_group) comm_group_end();
}

double *raw_slot(mg_slab_plan *p, size_t rec, size_t local_i) { return p->raw_dev + rec * p->local.size() + local_i; }

int add_record(mg_slab_plan *p, int node, int N, int steps, int final)
{
    p->records.push_back(mg_node_record{node, N, steps, 0.0});
    p->rec_final.push_back(final);
    return (int)p->records.size() - 1;
}

void run(mg_slab_plan *p)
{
    Context &c = ctx();
    size_t tok = 0;
    int at = 0;
    auto next = [&](double *v) {
        if (tok >= p->tokens.size()) return false;
        *v = p->tokens[tok++];
        return true;
    };
    const int step = p->con_step;
    c.defer_norms = true;
    bool came_back_up = false;  // the list has collapsed back to the finest level once

    for (;;) {
        double t;
        if (!next(&t)) break;
        const int node = (int)t;
        if (node == 2) break;
        if (c.last_error) { p->status = 10; break; }

        if (node == -1) {  // smooth + residual + sign flip + restrict, src/MG_solver_CPU.cpp:252-287
            if (at + 1 >= (int)p->sizes.size()) { p->status = 4; break; }
            // a second descent from the finest level would keep U (restart rule :252-257): not
            // implemented for slabs -- refuse instead of silently zeroing
            if (p->levels.size() == 1 && came_back_up) { p->status = 13; break; }
            const int hier = ++at;
            const int M = p->sizes[(size_t)hier];
            Level &cur = p->levels.back();
            Level nxt;
            nxt.N = M;
            nxt.hier = hier;
            nxt.collapsed = p->level_collapsed[(size_t)hier];
            if (!nxt.collapsed) {
                nxt.part = p->parts[(size_t)hier];
                nxt.halo = p->lp[(size_t)hier].halo;
            }
            alloc_level(p, nxt);
            {
                bool ok = true;
                for (const Local &l : nxt.loc)
                    if (!l.U || !l.F || !l.D) ok = false;
                if (!ok) { p->status = 14; break; }  // out of device memory
            }
            if ((int)p->records.size() >= (int)p->max_rec) { p->status = 11; break; }

            if (cur.collapsed) {
                const int rec = add_record(p, -1, cur.N, step, 1);
                for (size_t i = 0; i < p->local.size(); ++i) {
                    if (p->mixed)
                        mg_smooth_restrict_f32(cur.N, p->L, nullptr, (float *)cur.loc[i].U, (float *)cur.loc[i].F, step,
                                               raw_slot(p, (size_t)rec, i), M, (float *)nxt.loc[i].F);
                    else
                        mg_smooth_restrict(cur.N, p->L, nullptr, cur.loc[i].U, cur.loc[i].F, step, raw_slot(p, (size_t)rec, i), M,
                                           nxt.loc[i].F);
                }
                // levels N <= 64: the rest of this descent and its way back up in one launch
                // (every rank walks the same slice, so tokens and records stay in step)
                k::TailArgs ta;
                int node_level[k::TAIL_MAX_NODES];
                size_t tk = tok;
                if (scan_tail(p->tokens, &tk, p->sizes, at, step, M, p->L, &ta, node_level) &&
                    p->records.size() + (size_t)ta.n_nodes < p->max_rec) {
                    int rec_of[k::TAIL_MAX_NODES];
                    for (int i = 0; i < ta.n_nodes; ++i)
                        rec_of[i] = add_record(p, ta.nodes[i].type, ta.N[node_level[i]],
                                               ta.nodes[i].type == 0 ? 0 : ta.nodes[i].steps, 1);
                    for (size_t li = 0; li < p->local.size(); ++li) {
                        k::TailArgs mine = ta;
                        for (int i = 0; i < mine.n_nodes; ++i)  // error slots index raw_dev directly
                            if (mine.nodes[i].type != 0) mine.nodes[i].err_slot = (int)((size_t)rec_of[i] * p->local.size() + li);
                        mine.err_dev = p->raw_dev;
                        mine.gs_state = c.gs_state;
                        ProfScope ps("coarse_tail", M, 0.0);
                        if (p->mixed) {
                            k::TailArgsF f = tail_args_f32(mine);
                            f.F_top = (const float *)nxt.loc[li].F;
                            f.U_top = (float *)nxt.loc[li].U;
                            k::tail_launch_f32(c.stream, f);
                        } else {
                            mine.F_top = nxt.loc[li].F;
                            mine.U_top = nxt.loc[li].U;
                            k::tail_launch(c.stream, mine);
                        }
                    }
                    tok = tk;
                } else if (p->mixed && M <= k::TAIL_MAX_N) {
                    p->status = 15;  // mixed precision needs the coarse part of the file inside the tail kernel
                    break;
                }
            } else {
                const int rec = add_record(p, -1, cur.N, step, 0);
                const Partition cpart = nxt.collapsed ? induced_partition(cur.part, cur.N, M) : nxt.part;
                const LevelPlan &plan_cur = p->lp[(size_t)cur.hier];
                for (size_t i = 0; i < p->local.size(); ++i) {
                    const int r = p->local[i];
                    SlabFusion sf;
                    sf.fine_w = window_of(cur, r);
                    sf.fine_w.norm_lo = sf.fine_w.own_lo;  // the error counts owned rows only
                    sf.fine_w.norm_hi = sf.fine_w.own_hi;
                    // ... but the launch also updates the rows beyond them whose restricted residual the next
                    // level's launches will read (recomputed here instead of exchanged)
                    sf.fine_w.own_lo = plan_cur.dext[(size_t)r].lo;
                    sf.fine_w.own_hi = plan_cur.dext[(size_t)r].hi;
                    sf.M = M;
                    sf.Fc = nxt.loc[i].F;
                    // a collapsed next level is a full array on every rank: this rank writes its rows
                    sf.fc_w = nxt.collapsed ? RowWindow{0, M, cpart.lo[(size_t)r], cpart.hi[(size_t)r]} : window_of(nxt, r);
                    // U starts from zero on every descent (:252-257; a restart inside one file is
                    // not supported in slab mode), so the launch reads no U at all -- nor does it store one where
                    // the level's `1` launch recomputes it
                    sf.no_out = plan_cur.pre > 0;
                    if (p->mixed)
                        slab_smooth_f32(cur.N, p->L, nullptr, (float *)cur.loc[i].U, (const float *)cur.loc[i].F, step,
                                        raw_slot(p, (size_t)rec, i), sf);
                    else
                        slab_smooth(cur.N, p->L, nullptr, cur.loc[i].U, cur.loc[i].F, step, raw_slot(p, (size_t)rec, i), sf);
                }
                p->levels.push_back(nxt);
                Level &fine_lv = p->levels[p->levels.size() - 2], &next_lv = p->levels.back();
                // ONE group on the second stream: this level's U halo (read when the cycle comes back up through
                // this level: nobody waits for it before that), the next level's F halo where the schedule
                // exchanges it instead of recomputing it, and, at the collapse boundary, the all-gather of the
                // next level's F rows -- the latter two are on the critical path and are joined at once
                std::vector<GhostItem> items;
                if (plan_cur.xU > 0) items.push_back(GhostItem{&fine_lv, ARR_U, plan_cur.xU});
                const int xF = next_lv.collapsed ? 0 : p->lp[(size_t)next_lv.hier].xF;
                if (xF > 0) items.push_back(GhostItem{&next_lv, ARR_F, xF});
                exchange_ghosts(p, (size_t)fine_lv.hier, items, next_lv.collapsed ? &next_lv : nullptr, &cpart);
                if (xF > 0 || next_lv.collapsed) exchange_join(p, (size_t)fine_lv.hier);
                continue;
            }
            p->levels.push_back(nxt);
        } else if (node == 0) {  // :305-324
            double tol, opt;
            if (!next(&tol) || !next(&opt)) { p->status = 3; break; }
            Level &cur = p->levels.back();
            if (!cur.collapsed) { p->status = 12; break; }  // the exact solver runs on collapsed levels only
            if ((int)opt != 1) { p->status = 5; break; }
            if (p->mixed) { p->status = 15; break; }  // an exact solve outside the tail kernel: fp64 mode only
            add_record(p, 0, cur.N, 0, 1);
            for (size_t i = 0; i < p->local.size(); ++i) mg_doExactSolver(cur.N, p->L, cur.loc[i].U, cur.loc[i].F, tol, 1);
        } else if (node == 1) {  // prolong + add + smooth, :329-424
            --at;
            if (p->levels.size() < 2) { p->status = 6; break; }
            Level coarse = p->levels.back();
            p->levels.pop_back();
            Level &fine = p->levels.back();
            if ((int)p->records.size() >= (int)p->max_rec) { p->status = 11; break; }
            if (fine.collapsed) {
                const int rec = add_record(p, 1, fine.N, step, 1);
                for (size_t i = 0; i < p->local.size(); ++i) {
                    if (p->mixed)
                        mg_prolong_smooth_f32(coarse.N, (const float *)coarse.loc[i].U, fine.N, p->L, (const float *)fine.loc[i].U,
                                              (float *)fine.loc[i].D, (float *)fine.loc[i].F, step, raw_slot(p, (size_t)rec, i));
                    else
                        mg_prolong_smooth(coarse.N, coarse.loc[i].U, fine.N, p->L, fine.loc[i].U, fine.loc[i].D, fine.loc[i].F,
                                          step, raw_slot(p, (size_t)rec, i));
                    std::swap(fine.loc[i].U, fine.loc[i].D);
                }
            } else {
                const int rec = add_record(p, 1, fine.N, step, 0);
                // the fine level's U halo (where it was exchanged and not recomputed) must have arrived by now;
                // its F halo came with the descent, and the halo rows of the coarse U that this launch reads
                // were computed by this rank itself (ext, see slab_schedule)
                const size_t hier = (size_t)at;  // hierarchy index of the fine level
                exchange_join(p, hier);
                for (size_t i = 0; i < p->local.size(); ++i) {
                    const int r = p->local[i];
                    SlabFusion sf;
                    sf.fine_w = window_of(fine, r);
                    sf.fine_w.norm_lo = sf.fine_w.own_lo;  // the error counts owned rows only
                    sf.fine_w.norm_hi = sf.fine_w.own_hi;
                    sf.fine_w.own_lo = p->lp[hier].ext[(size_t)r].lo;
                    sf.fine_w.own_hi = p->lp[hier].ext[(size_t)r].hi;
                    sf.Nc = coarse.N;
                    sf.coarse = coarse.loc[i].U;
                    sf.coarse_w = coarse.collapsed ? RowWindow{0, coarse.N, 0, coarse.N} : window_of(coarse, r);
                    sf.pre = p->lp[hier].pre;
                    // the node that ends the first cycle of a refinement window on the finest level stores its rows of the
                    // fp64 iterate itself (exact widening in the store: no conversion pass; as mg_cycle.cpp does on one GPU)
                    if (p->mixed && p->refinements > 1 && p->refine_it == 0 && p->levels.size() == 1 && i < p->U64.size() &&
                        (tok >= p->tokens.size() || (int)p->tokens[tok] == 2)) {
                        sf.out_wide = p->U64[i];
                        p->top_widened = true;
                    }
                    if (p->mixed)
                        slab_smooth_f32(fine.N, p->L, sf.pre ? nullptr : (const float *)fine.loc[i].U, (float *)fine.loc[i].D,
                                        (const float *)fine.loc[i].F, step, raw_slot(p, (size_t)rec, i), sf);
                    else
                        slab_smooth(fine.N, p->L, sf.pre ? nullptr : fine.loc[i].U, fine.loc[i].D, fine.loc[i].F, step,
                                    raw_slot(p, (size_t)rec, i), sf);
                    std::swap(fine.loc[i].U, fine.loc[i].D);
                }
            }
            free_level(p, coarse);
            if (p->levels.size() == 1) came_back_up = true;
        }
    }
    for (size_t l = 0; l < p->pending.size(); ++l) exchange_join(p, l);  // a file that never came back up
    flush_norms();
    c.defer_norms = false;
}

}  // namespace

extern "C" {

int mg_slab_collect(mg_slab_plan *p, mg_cycle_result *out);

// host-only: the schedule of the distributed levels (see the head of this file).
//   ext[l][r]  rows a `1` node of level l updates on rank r: its owned rows plus what the next finer level's `1`
//              node will read of this level through the prolongation (rows orow[y], orow[y]+1 for every fine row y
//              that launch loads) -- computed redundantly instead of exchanged;
//   dext[l][r] rows a `-1` node updates: its owned rows plus the fine rows whose restricted residual the next level's
//              launches read beyond their owned rows (the next level's F halo), unless that costs more than max_pct
//              per cent of a slab, in which case the next level's F halo is exchanged (xF) and the growth starts anew;
//   xU[l]      rows of level l's U halo that have to travel because a `1` node reads them and the `-1` node did not
//              update them.
// Does level N of an nranks-slab plan run the recomputing node pair?  What decides is the size of a LAUNCH, not of the grid:
// the pair pays from ~8 M points per launch on (one GPU: N = 4096 gains 35 us, N = 2048 loses 4; 8 slabs of the 8192 level --
// 1024 x 8192 points each -- 90 against 101 us for the pair of launches, 8 slabs of the 4096 level 56 against 44:
// `MG_RECOMPUTE_MIN_N=... scripts/perf_slab.py 16384 8`, window 4.12 -> 4.05 ms) -- i.e. N * rows per slab >= min_n^2 / 2.
static bool slab_level_recomputes(int N, int nranks, int steps, int min_n)
{
    if (min_n <= 0 || !k::stream_recompute_supported(steps, steps) || N % 2 != 0) return false;
    return (double)N * (double)(N / (nranks > 0 ? nranks : 1)) >= 0.5 * (double)min_n * (double)min_n;
}

// ca_mode: 0 = exchange every halo (one group per level, the round-1 schedule), 1 = recompute F halos (default),
// 2 = recompute the U halos as well (no ghost exchange at all, only the collapse all-gather).
// recompute_min: levels at least this large run the node pair that neither stores nor re-reads the pre-smoothed U
// (0 = none: MG_SLAB_RECOMPUTE=0): their `1` launch reads F on grow(ext, 2*steps + 2) and no
// U at all, so such a level has no U halo to recompute or to exchange
static bool slab_schedule(const std::vector<int> &sizes, const std::vector<Partition> &parts, const std::vector<bool> &collapsed,
                          int nranks, int steps, int ca_mode, int max_pct, int recompute_min, std::vector<LevelPlan> *out)
{
    const size_t nl = sizes.size();
    size_t nd = 0;
    while (nd < nl && !collapsed[nd]) ++nd;  // levels 0 .. nd-1 are distributed
    const int H = steps + 2;  // input rows a launch loads beyond the rows it updates (Halo<S> + 1 spare)
    const int H2 = 2 * steps + 2;  // the same for the 2*steps levels of a recomputing `1` launch
    const size_t R = (size_t)nranks;
    std::vector<LevelPlan> lp(nl);
    for (size_t l = 0; l < nd; ++l) lp[l].pre = slab_level_recomputes(sizes[l], nranks, steps, recompute_min) ? steps : 0;
    // F rows the launches of level `P` read: the `-1` launch around dext, the `1` launch around ext
    auto f_rows = [&](const LevelPlan &P, size_t r, int n) { return unite(grow(P.dext[r], H, n), grow(P.ext[r], P.pre ? H2 : H, n)); };
    // owned rows and the rows of the way up, finest level first
    for (size_t l = 0; l < nd; ++l) {
        LevelPlan &P = lp[l];
        P.own.resize(R);
        P.ext.resize(R);
        P.dext.resize(R);
        P.fwr.resize(R);
        std::vector<int> owner;
        std::vector<double> wh, wl;
        if (l > 0) {
            owner.resize((size_t)sizes[l - 1]);
            wh.resize(owner.size());
            wl.resize(owner.size());
            build_prolongation_table(sizes[l], sizes[l - 1], 0, owner.data(), wh.data(), wl.data());
        }
        for (size_t r = 0; r < R; ++r) {
            P.own[r] = Span{parts[l].lo[r], parts[l].hi[r]};
            Span e = P.own[r];
            if (l > 0) {
                const int Nf = sizes[l - 1], Nc = sizes[l];
                const Span f = grow(lp[l - 1].ext[r], H, Nf);
                e.lo = std::min(e.lo, std::max(0, owner[(size_t)f.lo]));
                e.hi = std::max(e.hi, std::min(Nc, owner[(size_t)f.hi - 1] + 2));
            }
            P.ext[r] = e;
        }
    }
    // rows of the way down, coarsest distributed level first
    for (size_t l = nd; l-- > 0;) {
        LevelPlan &P = lp[l];
        const int N = sizes[l];
        int rows_min = 1 << 30;
        for (size_t r = 0; r < R; ++r) {
            P.dext[r] = P.own[r];
            if (ca_mode >= 2 && !P.pre) P.dext[r] = unite(P.dext[r], grow(P.ext[r], H, N));  // U halo of the way up recomputed as well
            rows_min = std::min(rows_min, P.own[r].hi - P.own[r].lo);
        }
        if (l + 1 < nd) {
            LevelPlan &C = lp[l + 1];
            const int M = sizes[l + 1];
            std::vector<int> lo((size_t)M);
            std::vector<double> w((size_t)M);
            build_restriction_table(N, M, lo.data(), w.data());
            std::vector<Span> cand(R);
            int growth = 0, needF = 0;
            for (size_t r = 0; r < R; ++r) {
                const Span need = f_rows(C, r, M);  // rows of the next level's F its launches read
                needF = std::max(needF, beyond(need, C.own[r]));
                cand[r] = P.dext[r];
                const int ra = std::max(1, need.lo), rb = std::min(M - 2, need.hi - 1);
                if (ra <= rb) cand[r] = unite(cand[r], Span{lo[(size_t)ra], lo[(size_t)rb] + 1});
                growth = std::max(growth, beyond(cand[r], P.own[r]));
            }
            C.needF = needF;
            if (ca_mode >= 1 && (long long)growth * 100 <= (long long)max_pct * rows_min) {
                P.dext = cand;  // the next level's F halo is recomputed here
                C.xF = 0;
            } else {
                C.xF = needF;   // ... or exchanged after this level's `-1` launch
            }
            // rows of the next level's F that each rank's `-1` launch writes
            for (size_t r = 0; r < R; ++r) {
                int first = -1, last = -2;
                for (int rc = 1; rc <= M - 2; ++rc) {
                    if (lo[(size_t)rc] >= P.dext[r].lo && lo[(size_t)rc] < P.dext[r].hi) {
                        if (first < 0) first = rc;
                        last = rc;
                    }
                }
                Span f = first >= 0 ? Span{first, last + 1} : Span{C.own[r].lo, C.own[r].lo};
                if (P.dext[r].lo == 0) f.lo = 0;      // the chunk that holds fine row 0 writes the coarse rim row 0,
                if (P.dext[r].hi == N) f.hi = M;      // the one with fine row N-1 the rim row M-1
                C.fwr[r] = f;
            }
        }
    }
    if (nd > 0) {
        LevelPlan &T = lp[0];
        for (size_t r = 0; r < R; ++r) {
            T.needF = std::max(T.needF, beyond(f_rows(T, r, sizes[0]), T.own[r]));
            T.fwr[r] = T.own[r];
        }
    }
    // U halos that have to travel, halo rows per level, and the check that a halo stays inside the neighbour's rows
    for (size_t l = 0; l < nd; ++l) {
        LevelPlan &P = lp[l];
        const int N = sizes[l];
        int rows_min = 1 << 30, halo = std::max(P.needF, P.xF);
        for (size_t r = 0; r < R; ++r) {
            rows_min = std::min(rows_min, P.own[r].hi - P.own[r].lo);
            const Span needU = grow(P.ext[r], H, N);
            if (!P.pre && !inside(needU, P.dext[r])) P.xU = std::max(P.xU, beyond(needU, P.own[r]));
            halo = std::max(halo, beyond(unite(P.dext[r], P.ext[r]), P.own[r]));
            halo = std::max(halo, beyond(P.fwr[r], P.own[r]));
        }
        halo = std::max(halo, P.xU);
        P.halo = halo;
        if (R > 1 && halo > rows_min) {
            fail(MG_ERR_UNSUPPORTED, "row-slab mode: level %d needs %d halo rows but its smallest slab has %d rows (raise collapse_N)",
                 N, halo, rows_min);
            return false;
        }
    }
    *out = lp;
    return true;
}

// levels from this size on run the recomputing node pair (fp64 plans; MG_SLAB_RECOMPUTE=0 switches it off)
static int slab_recompute_min(bool mixed)
{
    static const bool on = [] { const char *e = getenv("MG_SLAB_RECOMPUTE"); return !e || atoi(e) != 0; }();
    (void)mixed;  // fp32 slabs run the same pair
    return (!on || ctx().smoother == SMOOTHER_SIMPLE) ? 0 : recompute_min_n();
}

static int slab_ca_mode()
{
    const char *e = getenv("MG_SLAB_CA");
    const int m = e ? atoi(e) : 1;
    return m < 0 ? 0 : (m > 2 ? 2 : m);
}
static int slab_ca_pct()
{
    const char *e = getenv("MG_SLAB_CA_PCT");
    const int v = e ? atoi(e) : 10;
    return v < 0 ? 0 : v;
}

// host-only: the row ranges each rank owns on every level of the hierarchy a cycle file
// generates (N_max, halving down to N_min), and which levels are collapsed (replicated on every rank).
// out[(level*nranks + rank)*2 + {0,1}] = {lo, hi}; collapsed levels report {0, N} for rank 0
// and {0, 0} for the others.  Returns the number of levels.
int mg_slab_partition(int N_max, int N_min, int nranks, int collapse_N, int *out, int *collapsed_out)
{
    std::vector<int> sizes;
    for (int n = N_max; n >= N_min && n > 0; n /= 2) sizes.push_back(n);
    Partition cur = split_rows(N_max, nranks);
    bool collapsed = false;
    for (size_t l = 0; l < sizes.size(); ++l) {
        const int N = sizes[l];
        if (l > 0 && !collapsed) {
            Partition nx = induced_partition(cur, sizes[l - 1], N);
            if (N <= collapse_N || N % 2 != 0 || min_rows(nx) < 2 * MIN_HALF) collapsed = true;
            else cur = nx;
        }
        if (collapsed_out) collapsed_out[l] = collapsed ? 1 : 0;
        for (int r = 0; r < nranks; ++r) {
            if (out) {
                out[(l * (size_t)nranks + r) * 2 + 0] = collapsed ? 0 : cur.lo[(size_t)r];
                out[(l * (size_t)nranks + r) * 2 + 1] = collapsed ? (r == 0 ? N : 0) : cur.hi[(size_t)r];
            }
        }
    }
    return (int)sizes.size();
}

int mg_slab_ghost_rows(void) { return MIN_HALF; }

// host-only: pre_out[l] = sweeps the `1` launch of level l recomputes from zero instead of reading the level's U
// (0: the level stores and re-reads it), for the fp64 plan mg_slab_load would build
int mg_slab_recompute_levels_ranks(int N_max, int N_min, int steps, int nranks, int *pre_out)
{
    int nl = 0;
    const int min_n = slab_recompute_min(false);
    for (int n = N_max; n >= N_min && n > 0; n /= 2, ++nl)
        if (pre_out) pre_out[nl] = slab_level_recomputes(n, nranks, steps, min_n) ? steps : 0;
    return nl;
}
int mg_slab_recompute_levels(int N_max, int N_min, int steps, int *pre_out) { return mg_slab_recompute_levels_ranks(N_max, N_min, steps, 1, pre_out); }

// host-only: the schedule mg_slab_load derives for a hierarchy (`steps` sweeps per node; ca_mode / ca_pct < 0: the
// defaults resp. env MG_SLAB_CA / MG_SLAB_CA_PCT).  level_out[l*6 + ..] = {N, collapsed, halo, needF, xF, xU};
// rank_out[(l*nranks + r)*8 + ..] = {own.lo, own.hi, dext.lo, dext.hi, ext.lo, ext.hi, fwr.lo, fwr.hi} (zeros for
// collapsed levels).  Returns the number of levels, -1 when a halo would not fit its neighbour's slab.
int mg_slab_schedule(int N_max, int N_min, int nranks, int collapse_N, int steps, int ca_mode, int ca_pct, int *level_out,
                     int *rank_out)
{
    std::vector<int> sizes;
    for (int n = N_max; n >= N_min && n > 0; n /= 2) sizes.push_back(n);
    const size_t nl = sizes.size();
    std::vector<int> ranges(nl * (size_t)nranks * 2), coll(nl);
    mg_slab_partition(N_max, N_min, nranks, collapse_N, ranges.data(), coll.data());
    std::vector<Partition> parts(nl);
    std::vector<bool> collapsed(nl);
    for (size_t l = 0; l < nl; ++l) {
        collapsed[l] = coll[l] != 0;
        for (int r = 0; r < nranks; ++r) {
            parts[l].lo.push_back(ranges[(l * (size_t)nranks + r) * 2]);
            parts[l].hi.push_back(ranges[(l * (size_t)nranks + r) * 2 + 1]);
        }
    }
    std::vector<LevelPlan> lp;
    if (!slab_schedule(sizes, parts, collapsed, nranks, steps, ca_mode < 0 ? slab_ca_mode() : ca_mode,
                       ca_pct < 0 ? slab_ca_pct() : ca_pct, slab_recompute_min(false), &lp))
        return -1;
    for (size_t l = 0; l < nl; ++l) {
        if (level_out) {
            int *o = level_out + l * 6;
            o[0] = sizes[l];
            o[1] = collapsed[l] ? 1 : 0;
            o[2] = lp[l].halo;
            o[3] = lp[l].needF;
            o[4] = lp[l].xF;
            o[5] = lp[l].xU;
        }
        for (int r = 0; r < nranks && rank_out; ++r) {
            int *o = rank_out + (l * (size_t)nranks + r) * 8;
            for (int k = 0; k < 8; ++k) o[k] = 0;
            if (collapsed[l]) continue;
            const LevelPlan &P = lp[l];
            const Span v[4] = {P.own[(size_t)r], P.dext[(size_t)r], P.ext[(size_t)r], P.fwr[(size_t)r]};
            for (int k = 0; k < 4; ++k) {
                o[2 * k] = v[k].lo;
                o[2 * k + 1] = v[k].hi;
            }
        }
    }
    return (int)nl;
}

// rank >= 0: this process is that rank (RCCL communicator from mg_comm_init must exist when
// nranks > 1).  rank == -1: all nranks slabs live in this process (virtual ranks).
mg_slab_plan *mg_slab_load(const char *path, int nranks, int rank, int collapse_N)
{
    return mg_slab_load_flags(path, nranks, rank, collapse_N, 0);
}

// flags: MG_CYCLE_MIXED = fp32 fields for the whole cycle (source rounded once per slab, exact solver
// in fp64 inside the tail kernel, result widened); everything else as mg_slab_load
mg_slab_plan *mg_slab_load_flags(const char *path, int nranks, int rank, int collapse_N, int flags)
{
    if (!require_ready("mg_slab_load")) return nullptr;
    std::ifstream f(path);
    if (!f.is_open()) {
        fail(MG_ERR_CYCLE_FILE, "Cannot open file %s", path);
        return nullptr;
    }
    mg_slab_plan *p = new mg_slab_plan;
    p->path = path;
    p->mixed = (flags & MG_CYCLE_MIXED) != 0;
    p->elem = p->mixed ? sizeof(float) : sizeof(double);
    if (!(f >> p->L >> p->min_x >> p->min_y >> p->con_step >> p->con_N >> p->N_max >> p->N_min)) {
        fail(MG_ERR_CYCLE_FILE, "%s: malformed cycle structure header", path);
        delete p;
        return nullptr;
    }
    if (p->con_N != 1 || p->con_step < 1 || p->con_step > k::stream_max_steps() || p->N_max % 2 != 0) {
        fail(MG_ERR_UNSUPPORTED,
             "row-slab mode needs con_N = 1, a fixed con_step in 1..%d and an even N_max (got con_step=%d con_N=%d N_max=%d)",
             k::stream_max_steps(), p->con_step, p->con_N, p->N_max);
        delete p;
        return nullptr;
    }
    if (nranks < 1 || rank >= nranks || (rank >= 0 && nranks > 1 && (!comm_ready() || comm_size() != nranks || comm_rank() != rank))) {
        fail(MG_ERR_COMM, "mg_slab_load: rank %d of %d without a matching communicator (mg_comm_init)", rank, nranks);
        delete p;
        return nullptr;
    }
    std::string tk;
    while (f >> tk) {
        char *end = nullptr;
        const double v = strtod(tk.c_str(), &end);
        if (end == tk.c_str()) break;
        p->tokens.push_back(v);
    }
    for (int n = p->N_max; n >= p->N_min && n > 0; n /= 2) p->sizes.push_back(n);
    p->nranks = nranks;
    p->real = rank >= 0 && nranks > 1;
    if (rank >= 0) p->local.push_back(rank);
    else for (int r = 0; r < nranks; ++r) p->local.push_back(r);
    p->collapse_N = collapse_N;

    // partitions of the whole hierarchy
    const size_t nl = p->sizes.size();
    std::vector<int> ranges(nl * (size_t)nranks * 2), coll(nl);
    mg_slab_partition(p->N_max, p->N_min, nranks, collapse_N, ranges.data(), coll.data());
    p->parts.resize(nl);
    p->level_collapsed.resize(nl);
    for (size_t l = 0; l < nl; ++l) {
        p->level_collapsed[l] = coll[l] != 0;
        for (int r = 0; r < nranks; ++r) {
            p->parts[l].lo.push_back(ranges[(l * (size_t)nranks + r) * 2]);
            p->parts[l].hi.push_back(ranges[(l * (size_t)nranks + r) * 2 + 1]);
        }
    }
    if (p->level_collapsed[0] || min_rows(p->parts[0]) < 2 * MIN_HALF) {
        fail(MG_ERR_UNSUPPORTED, "row-slab mode: N_max=%d is too small for %d ranks", p->N_max, nranks);
        delete p;
        return nullptr;
    }
    if (!slab_schedule(p->sizes, p->parts, p->level_collapsed, nranks, p->con_step, slab_ca_mode(), slab_ca_pct(), slab_recompute_min(p->mixed), &p->lp)) {
        delete p;
        return nullptr;
    }
    p->poison = getenv("MG_SLAB_POISON") != nullptr;
    // second stream + event pairs for the exchanges (one pair per hierarchy index)
    if (!MG_HIP(hipStreamCreateWithFlags(&p->comm, hipStreamNonBlocking))) {
        delete p;
        return nullptr;
    }
    p->ev_ready.resize(nl);
    p->ev_done.resize(nl);
    p->pending.assign(nl, 0);
    for (size_t l = 0; l < nl; ++l) {
        (void)hipEventCreateWithFlags(&p->ev_ready[l], hipEventDisableTiming);
        (void)hipEventCreateWithFlags(&p->ev_done[l], hipEventDisableTiming);
    }
    size_t smoothing_nodes = 0;
    for (double t : p->tokens)
        if (t == -1.0 || t == 1.0 || t == 0.0) ++smoothing_nodes;
    p->max_rec = smoothing_nodes + 8;
    p->raw_dev = (double *)p->pool.get((p->max_rec + 1) * p->local.size() * sizeof(double));
    if (p->real) p->all_dev = (double *)p->pool.get((p->max_rec + 1) * (size_t)nranks * sizeof(double));

    // finest level: every rank evaluates getSource on its own window (:153)
    Level top;
    top.N = p->N_max;
    top.part = p->parts[0];
    top.halo = p->lp[0].halo;
    alloc_level(p, top);
    for (size_t i = 0; i < p->local.size(); ++i) {
        const RowWindow w = window_of(top, p->local[i]);
        const int lo = std::max(0, w.base), hi = std::min(top.N, w.base + w.rows);
        if (!p->mixed) {
            fill_source_rows(top.N, p->L, p->min_x, p->min_y, lo, hi, row_ptr(top.loc[i].F, w, top.N, lo));
            continue;
        }
        // mixed: the fp64 source rows are kept (the refinement forms its residual against them) and rounded
        // once into the fp32 field; a second fp64 window holds the widened result / the refinement's iterate
        double *f64 = (double *)p->pool.get((size_t)w.rows * top.N * sizeof(double));
        double *u64 = (double *)p->pool.get((size_t)w.rows * top.N * sizeof(double));
        if (!f64 || !u64) { mg_slab_destroy(p); return nullptr; }
        p->F64.push_back(f64);
        p->U64.push_back(u64);
        const size_t n = (size_t)(hi - lo) * top.N;
        fill_source_rows(top.N, p->L, p->min_x, p->min_y, lo, hi, row_ptr(f64, w, top.N, lo));
        k::convert_to_f32(ctx().stream, (float *)row_at(p, top.loc[i].F, w, top.N, lo), row_ptr(f64, w, top.N, lo), n);
    }
    p->levels.push_back(top);
    (void)hipEventCreate(&p->ev0);
    (void)hipEventCreate(&p->ev1);
    mg_sync();
    return p;
}

// one window on the engine's stream, no host synchronisation (see mg_cycle_enqueue)
int mg_slab_enqueue(mg_slab_plan *p)
{
    if (!require_ready("mg_slab_enqueue") || !p) return 1;
    Context &c = ctx();
    (void)hipEventRecord(p->ev0, c.stream);
    p->status = 0;
    p->U64_current = false;
    const int outer = p->mixed ? p->refinements : 1;
    for (int it = 0; it < outer && p->status == 0; ++it) {
        while (p->levels.size() > 1) {
            free_level(p, p->levels.back());
            p->levels.pop_back();
        }
        p->records.clear();  // those of the last cycle are reported
        p->rec_final.clear();
        c.active_pool = &p->pool;
        (void)hipMemsetAsync(p->raw_dev, 0, (p->max_rec + 1) * p->local.size() * sizeof(double), c.stream);
        Level &top = p->levels[0];
        if (it > 0 || p->F32_stale) {
            // mixed-precision refinement (mg_cycle_set_refinement on slabs): the fp32 source of this cycle is
            // the fp64 residual of the fp64 iterate (one halo row of it comes from the neighbours), rounded;
            // its own halo rows are exchanged like any F.  A new window first restores the rounded source.
            if (it > 0) {  // the 5-point star: one row
                exchange_ghosts(p, 0, {GhostItem{&top, ARR_U, 1, &p->U64, sizeof(double)}});
                exchange_join(p, 0);
            }
            const bool own_array = it > 0 && p->F32_res.size() == p->local.size();
            if (own_array) {  // the correction cycle's source lives in its own windows: the finest level points at them
                p->F32_src.clear();
                for (size_t i = 0; i < p->local.size(); ++i) {
                    p->F32_src.push_back(top.loc[i].F);
                    top.loc[i].F = p->F32_res[i];
                }
            }
            for (size_t i = 0; i < p->local.size(); ++i) {
                const RowWindow w = window_of(top, p->local[i]);
                if (it > 0) {
                    const double dx = p->L / (double)(top.N - 1);
                    k::refine_residual_rows(c.stream, top.N, 1.0 / (dx * dx), p->U64[i], p->F64[i], (float *)top.loc[i].F, w,
                                            p->refine_raw + (size_t)(it - 1) * p->local.size() + i);
                } else {
                    const int lo = std::max(0, w.base), hi = std::min(top.N, w.base + w.rows);
                    k::convert_to_f32(c.stream, (float *)row_at(p, top.loc[i].F, w, top.N, lo), row_ptr(p->F64[i], w, top.N, lo),
                                      (size_t)(hi - lo) * top.N);
                }
            }
            if (it > 0 && p->lp[0].needF > 0) {  // every halo row of the source the level's launches read
                exchange_ghosts(p, 0, {GhostItem{&top, ARR_F, p->lp[0].needF}});
                exchange_join(p, 0);
            }
            p->F32_stale = it > 0 && !own_array;
        }
        p->refine_it = it;
        p->top_widened = false;
        run(p);
        if (!p->F32_src.empty()) {  // back to the rounded source of the first cycle
            Level &t0 = p->levels[0];
            for (size_t i = 0; i < p->local.size(); ++i) t0.loc[i].F = p->F32_src[i];
            p->F32_src.clear();
        }
        if (p->mixed && outer > 1 && p->status == 0) {
            // fp64 correction on the owned rows: U64 = (double)e on the first cycle, U64 += (double)e afterwards
            Level &top = p->levels[0];  // (run() grows the level vector: the reference above is gone)
            for (size_t i = 0; i < p->local.size(); ++i) {
                const RowWindow w = window_of(top, p->local[i]);
                const size_t n = (size_t)(w.own_hi - w.own_lo) * top.N;
                double *u64 = row_ptr(p->U64[i], w, top.N, w.own_lo);
                const float *e = (const float *)row_at(p, top.loc[i].U, w, top.N, w.own_lo);
                if (it == 0) {
                    if (!p->top_widened) k::convert_to_f64(c.stream, u64, e, n);
                } else {
                    k::add_widened(c.stream, u64, e, n);
                }
            }
            p->U64_current = true;
        }
    }
    (void)hipEventRecord(p->ev1, c.stream);
    c.active_pool = nullptr;
    return p->status;
}

int mg_slab_execute(mg_slab_plan *p, mg_cycle_result *out)
{
    if (!require_ready("mg_slab_execute") || !p) return 1;
    mg_sync();
    const auto t0 = std::chrono::steady_clock::now();
    mg_slab_enqueue(p);
    mg_sync();
    const auto t1 = std::chrono::steady_clock::now();
    const int status = mg_slab_collect(p, out);
    out->time_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    return status;
}

// wait for the enqueued windows and report the last one
int mg_slab_collect(mg_slab_plan *p, mg_cycle_result *out)
{
    if (!require_ready("mg_slab_collect") || !p) return 1;
    Context &c = ctx();
    memset(out, 0, sizeof *out);
    mg_sync();
    float dev_ms = 0.f;
    if (hipEventElapsedTime(&dev_ms, p->ev0, p->ev1) != hipSuccess) (void)hipGetLastError();

    // outside the window: the analytic error (:434-445) and the smoothing errors, combined
    // over the slabs in rank order
    Level &top = p->levels[0];
    const size_t nloc = p->local.size(), nrec = p->records.size();
    for (size_t i = 0; i < nloc && p->want_error; ++i) {
        const RowWindow w = window_of(top, p->local[i]);
        const double *U = top.loc[i].U;
        if (p->mixed) {  // widen the whole window (exact) unless the refinement kept U64 up to date
            if (!p->U64_current) k::convert_to_f64(c.stream, p->U64[i], (const float *)top.loc[i].U, (size_t)w.rows * top.N);
            U = p->U64[i];
        }
        k::analytic_error_rows(c.stream, top.N, p->L, U, w, p->min_x, p->min_y, raw_slot(p, p->max_rec, i));
    }
    std::vector<double> raw((p->max_rec + 1) * (size_t)p->nranks, 0.0);  // [slot][global rank]
    if (p->real) {
        comm_allgather(p->raw_dev, p->all_dev, p->max_rec + 1);  // nloc == 1: [slot] per rank (engine's stream: everything is joined)
        std::vector<double> tmp((p->max_rec + 1) * (size_t)p->nranks);
        mg_download(tmp.data(), p->all_dev, tmp.size());
        for (int r = 0; r < p->nranks; ++r)
            for (size_t s = 0; s <= p->max_rec; ++s) raw[s * (size_t)p->nranks + r] = tmp[(size_t)r * (p->max_rec + 1) + s];
    } else {
        mg_download(raw.data(), p->raw_dev, raw.size());
    }
    for (size_t k2 = 0; k2 < nrec; ++k2) {
        mg_node_record &rec = p->records[k2];
        if (rec.node == 0) continue;
        if (p->rec_final[k2]) {
            rec.error = raw[k2 * (size_t)p->nranks + 0];
            continue;
        }
        double s = 0.0;
        for (int r = 0; r < p->nranks; ++r) s += raw[k2 * (size_t)p->nranks + r];
        double e = s + s;  // :621-622
        e = e / rec.N / rec.N;
        rec.error = e;
    }
    p->refine_err.clear();
    if (p->mixed && p->refinements > 1 && p->refine_raw) {
        const size_t k = (size_t)p->refinements - 1;
        std::vector<double> rr(k * (size_t)p->nranks, 0.0);  // [iterate][global rank]
        if (p->real) {
            comm_allgather(p->refine_raw, p->refine_all, k);
            std::vector<double> tmp(k * (size_t)p->nranks);
            mg_download(tmp.data(), p->refine_all, tmp.size());
            for (int r = 0; r < p->nranks; ++r)
                for (size_t j = 0; j < k; ++j) rr[j * (size_t)p->nranks + r] = tmp[(size_t)r * k + j];
        } else {
            mg_download(rr.data(), p->refine_raw, rr.size());
        }
        for (size_t j = 0; j < k; ++j) {
            double sum = 0.0;
            for (int r = 0; r < p->nranks; ++r) sum += rr[j * (size_t)p->nranks + r];
            double e = sum + sum;  // :621-622
            e = e / top.N / top.N;
            p->refine_err.push_back(e);
        }
    }
    double a = 0.0;
    for (int r = 0; r < p->nranks; ++r) a += raw[p->max_rec * (size_t)p->nranks + r];

    out->status = p->status ? p->status : (c.last_error ? 10 : 0);
    out->N = top.N;
    out->U_dev = nullptr;
    out->mg_error = a / (double)(top.N * top.N);
    out->time_ms = 0.0;
    out->device_ms = dev_ms;
    out->n_records = (int)nrec;
    out->records = p->records.data();
    out->report = "";
    return out->status;
}

// the owned rows of this process's slabs of the finest U, into a full N x N host array
int mg_slab_gather_U(mg_slab_plan *p, double *host_full)
{
    if (!require_ready("mg_slab_gather_U") || !p) return 1;
    Level &top = p->levels[0];
    for (size_t i = 0; i < p->local.size(); ++i) {
        const RowWindow w = window_of(top, p->local[i]);
        double *U = top.loc[i].U;
        if (p->mixed) {
            if (!p->U64_current) k::convert_to_f64(ctx().stream, p->U64[i], (const float *)top.loc[i].U, (size_t)w.rows * top.N);
            U = p->U64[i];
        }
        mg_download(host_full + (size_t)w.own_lo * top.N, row_ptr(U, w, top.N, w.own_lo), (size_t)(w.own_hi - w.own_lo) * top.N);
    }
    return 0;
}

// mixed-precision slabs: `cycles` fp32 runs of the file per window joined by the fp64 residual of the fp64
// iterate and an fp64 correction (see mg_cycle_set_refinement); per extra cycle one ghost exchange of the
// fp64 iterate and one of the new fp32 source, both on the finest level
int mg_slab_set_refinement(mg_slab_plan *p, int cycles)
{
    if (!require_ready("mg_slab_set_refinement") || !p) return 1;
    if (!p->mixed || cycles < 1 || cycles > 64) {
        fail(MG_ERR_ARG, "mg_slab_set_refinement: needs a MG_CYCLE_MIXED plan and 1..64 cycles (got %d)", cycles);
        return 1;
    }
    if (!p->refine_raw) {
        p->refine_raw = (double *)p->pool.get(64 * p->local.size() * sizeof(double));
        if (p->real) p->refine_all = (double *)p->pool.get(64 * (size_t)p->nranks * sizeof(double));
        if (!p->refine_raw) return 1;
    }
    if (cycles > 1 && p->F32_res.empty() && !p->levels.empty()) {
        // windows of the finest level's geometry for the correction cycles' source (all or nothing: without them the
        // source array is reused and re-rounded at the start of every window)
        const Level &top = p->levels[0];
        for (size_t i = 0; i < p->local.size(); ++i) {
            const RowWindow w = window_of(top, p->local[i]);
            double *b = (double *)p->pool.get((size_t)w.rows * top.N * sizeof(float));
            if (!b) {
                for (double *q : p->F32_res) p->pool.put(q);
                p->F32_res.clear();
                break;
            }
            p->F32_res.push_back(b);
        }
    }
    p->refinements = cycles;
    return 0;
}

int mg_slab_refinement_errors(mg_slab_plan *p, double *out, int cap)
{
    if (!p || !out) return 0;
    int n = 0;
    for (; n < (int)p->refine_err.size() && n < cap; ++n) out[n] = p->refine_err[(size_t)n];
    return n;
}

void mg_slab_want_error(mg_slab_plan *p, int on)
{
    if (p) p->want_error = on != 0;
}

void mg_slab_destroy(mg_slab_plan *p)
{
    if (!p) return;
    if (ctx().ready) (void)hipStreamSynchronize(ctx().stream);
    for (Level &lv : p->levels) free_level(p, lv);
    p->levels.clear();
    if (p->raw_dev) p->pool.put(p->raw_dev);
    if (p->all_dev) p->pool.put(p->all_dev);
    for (double *b : p->U64) p->pool.put(b);
    for (double *b : p->F64) p->pool.put(b);
    for (double *b : p->F32_res) p->pool.put(b);
    if (p->refine_raw) p->pool.put(p->refine_raw);
    if (p->refine_all) p->pool.put(p->refine_all);
    if (p->ev0) (void)hipEventDestroy(p->ev0);
    if (p->ev1) (void)hipEventDestroy(p->ev1);
    if (p->comm) {
        (void)hipStreamSynchronize(p->comm);
        (void)hipStreamDestroy(p->comm);
    }
    for (hipEvent_t e : p->ev_ready) (void)hipEventDestroy(e);
    for (hipEvent_t e : p->ev_done) (void)hipEventDestroy(e);
    p->pool.trim();
    delete p;
}

}  // extern "C"
