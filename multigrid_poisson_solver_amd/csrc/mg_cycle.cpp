// mg_cycle.cpp -- the cycle-structure-file driver and the level stack.
//
// Keeps the reference's driver design (BASELINE.json north_star): the multigrid cycle
// is not hard-coded but interpreted, node by node, from a text file
// (README.md:43-128), and the per-level grids U/F/D live in a doubly linked list that
// is pushed on restriction and popped on prolongation (src/linkedlist.{h,cpp}).
// What changes: the list hands out DEVICE arrays from a plan-private pool, the driver's
// own host loops over U/F/D (memset :213,:256; sign flip :277-280; tempU :353,:371;
// final error :434-445) become engine calls, and the smoothing errors stay on the
// device until the window ends so the window has no host synchronisation.
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "mg_internal.h"

namespace mg {
namespace {

// ---------------------------------------------------------------------------
// LevelList: the reference's LinkedList (src/linkedlist.h:32-60) on device arrays
// ---------------------------------------------------------------------------
struct LevelNode {  // src/linkedlist.h:4-30
    int N = 0;
    double *U = nullptr, *F = nullptr, *D = nullptr;
    LevelNode *nextNode = nullptr, *prevNode = nullptr;
    int step = 0;
    double smoothingError = 0.0;
    // > 0: U holds nothing yet -- it is `pending_pre` Jacobi sweeps from zero on this F, which the fused `-1` node did
    // not store (smooth_restrict_no_out); the fused `1` node recomputes it in flight, anyone else calls ensure_U first
    int pending_pre = 0;
    // how many `-1` nodes have left this level since it was pushed (the dataflow trace gives every re-descent arrays of
    // its own: mg_cycle_plan::sched)
    int descents = 0;
};

class LevelList {
public:
    // elem_bytes: 8 (fp64 fields) or 4 (the fp32 fields of the mixed-precision mode; the typed
    // double* members then only carry the address)
    explicit LevelList(Pool *pool, size_t elem_bytes = sizeof(double)) : pool_(pool), elem_bytes_(elem_bytes) {}
    ~LevelList() { clear(); }
    void clear()
    {
        while (lastNode_) Remove_back();
        init_ = 1;
    }
    void Push_back(int n)  // src/linkedlist.cpp:28-44 (+ ListNode::ListNode :7-16)
    {
        LevelNode *nd = new LevelNode;
        const size_t bytes = (size_t)n * n * elem_bytes_;
        nd->N = n;
        nd->U = (double *)pool_->get(bytes);
        nd->F = (double *)pool_->get(bytes);
        nd->D = (double *)pool_->get(bytes);
        if (!firstNode_) {
            firstNode_ = lastNode_ = nd;
            return;
        }
        nd->prevNode = lastNode_;
        lastNode_->nextNode = nd;
        lastNode_ = nd;
    }
    void Remove_back()  // src/linkedlist.cpp:46-69
    {
        LevelNode *nd = lastNode_;
        if (!nd) return;
        pool_->put(nd->U);
        pool_->put(nd->F);
        pool_->put(nd->D);
        if (firstNode_ == lastNode_) {
            firstNode_ = lastNode_ = nullptr;
        } else {
            lastNode_ = nd->prevNode;
            lastNode_->nextNode = nullptr;
            if (firstNode_ == lastNode_) init_ = 0;
        }
        delete nd;
    }
    LevelNode *last() { return lastNode_; }
    LevelNode *first() { return firstNode_; }
    int depth() const
    {
        int d = 0;
        for (LevelNode *p = firstNode_; p; p = p->nextNode) ++d;
        return d;
    }
    int Get_N() const { return lastNode_->N; }
    int Get_prev_N() const { return lastNode_->prevNode->N; }
    int Get_init() const { return init_; }
    void Set_init(int r) { init_ = r; }
    bool Is_firstNode() const { return firstNode_ == lastNode_ && firstNode_ != nullptr; }

private:
    Pool *pool_;
    size_t elem_bytes_;
    LevelNode *firstNode_ = nullptr, *lastNode_ = nullptr;
    int init_ = 1;  // 1 until the list has collapsed back to the first node once
};

struct Text {
    std::string s;
    void printf(const char *fmt, ...)
    {
        char tmp[256];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(tmp, sizeof tmp, fmt, ap);
        va_end(ap);
        s += tmp;
    }
};

// report entries: literal text or a smoothing block whose error is filled in later
struct ReportItem {
    std::string text;
    int record = -1;  // >= 0: "~Smoothing~" block of records[record]
};

}  // namespace
}  // namespace mg

using namespace mg;

struct mg_cycle_plan {
    int flags = 0;
    std::string path;
    // header, src/MG_solver_CPU.cpp:103-109
    double L = 1.0, min_x = 0.0, min_y = 0.0;
    int con_step = 0, con_N = 0, N_max = 0, N_min = 0;
    std::vector<int> sizes;          // N_array :111-146
    std::vector<double> tokens;      // everything after the header
    Pool pool;                       // plan-private arena (stable addresses for graph replay)
    LevelList *levels = nullptr;
    double *F_finest = nullptr;      // getSource(N_max), evaluated once at load (:153)
    double *F64 = nullptr, *U64 = nullptr;  // mixed mode: the fp64 source and the widened result
    int refinements = 1;             // mixed mode: fp32 cycles per window (fp64 residual + correction between them)
    int refine_it = 0;               // ... and which of them is running
    float *F32_res = nullptr;        // the fp32 source of the correction cycles (the rounded residual of the fp64 iterate): an array
                                     // of its own, so that the rounded source of the first cycle never has to be made again
    bool F32_stale = false;          // the finest fp32 F holds a residual, not the rounded source
    double *refine_err_dev = nullptr;
    std::vector<double> refine_err;
    double *err_dev = nullptr;       // one slot per smoothing record
    size_t err_cap = 0;
    std::vector<mg_node_record> records;
    std::vector<ReportItem> report_items;
    std::string report;
    double *final_U = nullptr;
    int final_N = 0;
    // graph replay
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    bool graph_ready = false, graph_failed = false;
    int warm_runs = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int last_status = 0;
    bool last_replayed = false;      // the last window was a hipGraph replay
    // ---- batched breadth-first schedule ------------------------------------------------------------------------------
    // The reference zeroes a level's U before every pre-smoothing except the restart case (src/MG_solver_CPU.cpp:252-257),
    // so a `-1` node reads nothing but its level's F.  When a cycle file comes back up to a level and descends from it
    // AGAIN (the second half of every W-cycle visit: ... 1 -1 ...), that second sub-cycle depends on the level's F alone
    // -- not on anything the first sub-cycle produced (whose result the zero fill discards; only its smoothing errors
    // are ever looked at).  So the node program of a fixed-step file is a DATAFLOW GRAPH in which all 2^l visits of level
    // l of a W-cycle are independent of one another.  build_schedule() traces the file ONCE with the interpreter below
    // (every fused node described instead of launched, every visit on arrays of its own), orders the nodes by their depth
    // in that graph and merges the nodes of one depth and one shape into ONE launch with a batch dimension: a W-cycle
    // becomes as many launches as a V-cycle (2 per level + one coarse-tail launch of 64 workgroups at 8192^2).  Every
    // node still runs, on the same inputs, with the same expressions: same bits; records and report stay in file order.
    std::vector<NodeOp> ops;             // the traced nodes, file order
    std::vector<k::TailArgs> tails;      // coarse-tail slices among them (NodeOp::tail)
    struct Group {
        int depth = 0, first = 0;        // dataflow depth, first op (file order)
        std::vector<int> members;        // ops of this launch
        size_t items_at = 0;             // where its instance table starts in sched_items (bytes)
        std::vector<double *> errs;      // per instance: device slot of its smoothing error
    };
    std::vector<Group> sched;
    void *sched_items = nullptr;         // device: the instance tables of all groups
    bool sched_ready = false, sched_tried = false;
    int sched_max_batch = 0;
    int sched_smoother = 0;              // the smoother setting the trace ran under (another one runs the file node by node)
    double *sched_U = nullptr;           // where the schedule leaves the result
    int sched_N = 0;
    int *gs_slots = nullptr;             // [2 * tails] exact-solver state per coarse-tail instance
    int last_tail = -1;                  // the tail whose state mg_lastExactSolverIterations() reports (last in file order)
};

namespace mg {
// Shared by the single-GPU and the row-slab driver: starting at tokens[*tok], collect the slice
// of the node stream that stays on the level of size top_N (<= TAIL_MAX_N) and below, up to but
// not including the `1` that leaves it, into a TailArgs (levels, spacings, transfer tables; the
// caller fills F_top/U_top/err_dev/gs_state and the error slots).  node_level[i] = tail level
// node i works on.  Returns false, consuming nothing, when the slice has another shape.
// the same slice with fp32 fields: spacings and transfer weights rounded to fp32 (the exact solver
// keeps its fp64 spacings); F_top / U_top / err_dev / gs_state are the caller's to set
k::TailArgsF tail_args_f32(const k::TailArgs &a)
{
    k::TailArgsF f;
    memset(&f, 0, sizeof f);
    f.n_levels = a.n_levels;
    f.n_nodes = a.n_nodes;
    for (int i = 0; i < a.n_nodes; ++i) f.nodes[i] = a.nodes[i];
    for (int l = 0; l < a.n_levels; ++l) {
        f.N[l] = a.N[l];
        f.dx2[l] = (float)a.dx2[l];
        f.inv[l] = (float)a.inv[l];
        f.gs_h2[l] = a.gs_h2[l];
        f.gs_inv[l] = a.gs_inv[l];
        if (l + 1 < a.n_levels) {
            const RestrictTable &rt = restrict_table(a.N[l], a.N[l + 1]);
            const ProlongTable &pt = prolong_table(a.N[l + 1], a.N[l]);
            f.r_lo[l] = rt.lo;
            f.r_w[l] = rt.w_f;
            f.p_orow[l] = pt.owner_row;
            f.p_ocol[l] = pt.owner_col;
            f.p_rhi[l] = pt.row_hi_f;
            f.p_rlo[l] = pt.row_lo_f;
            f.p_chi[l] = pt.col_hi_f;
            f.p_clo[l] = pt.col_lo_f;
            f.c_dx[l] = (float)pt.c_dx;
        }
    }
    f.err_dev = a.err_dev;
    f.gs_state = a.gs_state;
    f.trace = a.trace;
    return f;
}

bool scan_tail(const std::vector<double> &tokens, size_t *tok_io, const std::vector<int> &sizes, int at0, int con_step,
               int top_N, double L, k::TailArgs *out, int *node_level)
{
    static const bool disabled = getenv("MG_NO_TAIL") != nullptr;
    if (disabled || con_step < 1 || top_N > k::TAIL_MAX_N || top_N < 3) return false;
    k::TailArgs &a = *out;
    memset(&a, 0, sizeof a);
    a.N[0] = top_N;
    int depth = 0, max_depth = 0, at = at0, n_nodes = 0;
    size_t tok = *tok_io;
    for (;;) {
        if (tok >= tokens.size()) return false;
        const int node = (int)tokens[tok++];
        if (node == 1 && depth == 0) {  // leaves the tail: not ours
            --tok;
            break;
        }
        if (n_nodes >= k::TAIL_MAX_NODES) return false;
        k::TailNode nd;
        memset(&nd, 0, sizeof nd);
        nd.type = node;
        nd.err_slot = -1;
        if (node == -1) {
            if (at + 1 >= (int)sizes.size() || depth + 1 >= k::TAIL_MAX_LEVELS) return false;
            nd.steps = con_step;
            node_level[n_nodes] = depth;
            ++depth;
            a.N[depth] = sizes[(size_t)++at];
            if (a.N[depth] < 3) return false;
            if (depth > max_depth) max_depth = depth;
        } else if (node == 0) {
            if (tok + 1 >= tokens.size()) return false;
            nd.tol = tokens[tok++];
            if ((int)tokens[tok++] != 1) return false;
            node_level[n_nodes] = depth;
        } else if (node == 1) {
            nd.steps = con_step;
            --depth;
            --at;
            node_level[n_nodes] = depth;  // the level that is smoothed
        } else {
            return false;  // 2 / end of file / anything else inside the slice
        }
        a.nodes[n_nodes++] = nd;
    }
    if (n_nodes == 0 || at != at0) return false;
    a.n_nodes = n_nodes;
    a.n_levels = max_depth + 1;
    if (!k::tail_fits(a)) return false;
    for (int l = 0; l < a.n_levels; ++l) {
        const double dx = L / (double)(a.N[l] - 1);
        a.dx2[l] = dx * dx;
        a.inv[l] = 1.0 / a.dx2[l];
        a.gs_h2[l] = a.dx2[l];
        a.gs_inv[l] = a.inv[l];
        if (l + 1 < a.n_levels) {
            const RestrictTable &rt = restrict_table(a.N[l], a.N[l + 1]);
            const ProlongTable &pt = prolong_table(a.N[l + 1], a.N[l]);
            if (!rt.lo || !pt.owner_row) return false;
            a.r_lo[l] = rt.lo;
            a.r_w[l] = rt.w;
            a.p_orow[l] = pt.owner_row;
            a.p_ocol[l] = pt.owner_col;
            a.p_rhi[l] = pt.row_hi;
            a.p_rlo[l] = pt.row_lo;
            a.p_chi[l] = pt.col_hi;
            a.p_clo[l] = pt.col_lo;
            a.c_dx[l] = pt.c_dx;
        }
    }
    *tok_io = tok;
    return true;
}
}  // namespace mg

namespace {

const double TRIGGER = 0.01;  // src/MG_solver_CPU.cpp:99

struct Exec {
    mg_cycle_plan *p;
    Context &c;
    size_t tok = 0;
    int at = 0;  // len_flag
    int status = 0;
    bool capturing = false;
    bool widened = false;  // mixed mode: the last node stored its result in fp64 (plan->U64) itself
    // dataflow trace (build_schedule): nodes are described, not launched; every visit of a level gets arrays of its own;
    // a node that is not one fused launch (restart, trigger mode, a stand-alone exact solve, 0 steps ...) ends the attempt
    bool tracing = false;
    bool untraceable = false;

    bool next(double *v)
    {
        if (tok >= p->tokens.size()) return false;
        *v = p->tokens[tok++];
        return true;
    }
    bool next_int(int *v)
    {
        double d;
        if (!next(&d)) return false;
        *v = (int)d;
        return true;
    }
};

int add_record(mg_cycle_plan *p, int node, int N, int steps)
{
    mg_node_record r{node, N, steps, 0.0};
    p->records.push_back(r);
    return (int)p->records.size() - 1;
}

void report_text(mg_cycle_plan *p, const char *t)
{
    if (!(p->flags & MG_CYCLE_REPORT)) return;
    ReportItem it;
    it.text = t;
    p->report_items.push_back(it);
}
void report_smoothing(mg_cycle_plan *p, int record)
{
    if (!(p->flags & MG_CYCLE_REPORT)) return;
    ReportItem it;
    it.record = record;
    p->report_items.push_back(it);
}

double *error_slot(mg_cycle_plan *p, int record) { return p->err_dev + record; }

// smoothing with the error trigger, src/MG_solver_CPU.cpp:194-230 / :376-402: one sweep
// at a time, each followed by a host-visible error -- inherently synchronous
int trigger_smoothing(Exec &x, LevelNode *lv)
{
    double slope = TRIGGER + 1.0, before = 0.0;
    lv->step = 0;
    while (slope > TRIGGER) {
        mg_doSmoothing(lv->N, x.p->L, lv->U, lv->F, 1, &lv->smoothingError);
        if (x.c.last_error) return lv->step;
        lv->step += 1;
        if (lv->step == 1) {
            before = lv->smoothingError;
            continue;
        }
        slope = std::fabs(lv->smoothingError - before);
        before = lv->smoothingError;
    }
    return lv->step;
}

// The sweeps of the `1` node that will come back up to the level a `-1` node is about to leave (x.tok stands behind that
// node's operands): the file's fixed step count, or -- per-node step counts, con_step = 0 -- the operand of the matching
// `1` further down the stream (src/MG_solver_CPU.cpp:331-344).  -1: none (the file ends below, trigger mode).
int matching_up_step(const Exec &x)
{
    const mg_cycle_plan *p = x.p;
    if (p->con_step > 0) return p->con_step;
    if (p->con_step != 0) return -1;
    int depth = 1;
    for (size_t tok = x.tok; tok < p->tokens.size();) {
        const int node = (int)p->tokens[tok++];
        if (node == 2) break;
        if (node == -1) {
            if (tok >= p->tokens.size()) break;
            const int s = (int)p->tokens[tok++];
            if (p->con_N == 0) ++tok;
            if (s != 0) ++depth;          // (a 0-step `-1` node does nothing: :241-243)
        } else if (node == 0) {
            tok += 2;
        } else if (node == 1) {
            if (tok >= p->tokens.size()) break;
            const int s = (int)p->tokens[tok++];
            if (--depth == 0) return s;
        }
    }
    return -1;
}

// materialise a pre-smoothed field that its `-1` node left to be recomputed (the rare consumer that is not the fused
// `1` node: a 0-step `1` node, a file that ends on the way down)
void ensure_U(mg_cycle_plan *p, LevelNode *lv)
{
    if (!lv || lv->pending_pre <= 0) return;
    if (p->flags & MG_CYCLE_MIXED) return;  // fp32 fields: every consumer is a fused `1` node (run_nodes turns anything else into status 15)
    mg_smooth_pp(lv->N, p->L, nullptr, lv->U, lv->F, lv->pending_pre, nullptr, nullptr, -1);
    lv->pending_pre = 0;
}

// fixed-step smoothing of the last level, result in lv->U.  zero_start: the driver's
// memset(U,0) (:256) is folded into the first sweep.  want_D: also produce -residual in
// lv->D (getResidual :268 and the sign flip :277-280 folded into the last sweep).
void smooth_level(Exec &x, LevelNode *lv, int step, bool zero_start, bool want_D, int record)
{
    mg_cycle_plan *p = x.p;
    const size_t n = (size_t)lv->N * lv->N;
    if (p->flags & MG_CYCLE_FUSED) {
        if (zero_start) {
            mg_smooth_pp(lv->N, p->L, nullptr, lv->U, lv->F, step, error_slot(p, record), want_D ? lv->D : nullptr, -1);
        } else if (!want_D) {
            // post-smoothing: D of this level is dead (consumed by the restriction on
            // the way down), use it as the ping-pong partner and swap
            mg_smooth_pp(lv->N, p->L, lv->U, lv->D, lv->F, step, error_slot(p, record), nullptr, -1);
            std::swap(lv->U, lv->D);
        } else {
            double *tmp = (double *)p->pool.get(n * sizeof(double));
            mg_smooth_pp(lv->N, p->L, lv->U, tmp, lv->F, step, error_slot(p, record), lv->D, -1);
            std::swap(lv->U, tmp);
            p->pool.put(tmp);
        }
        return;
    }
    // literal operator sequence of the reference
    if (zero_start) mg_fill_zero(lv->U, n);
    {
        double *tmp = (double *)p->pool.get(n * sizeof(double));
        mg_smooth_pp(lv->N, p->L, lv->U, tmp, lv->F, step, error_slot(p, record), nullptr, +1);
        mg_copy(lv->U, tmp, n);
        p->pool.put(tmp);
    }
    if (want_D) {
        mg_getResidual(lv->N, p->L, lv->U, lv->F, lv->D);
        mg_negate(lv->N, lv->D);
    }
}

// The coarse tail: once a descent reaches a level with N <= TAIL_MAX_N, the slice of the node
// stream that stays on that level and below (up to, not including, the `1` that leaves it)
// runs as ONE launch with all those levels in LDS (mg_tail.hip).  Returns false -- and consumes
// nothing -- whenever the slice is not of the supported shape; the nodes are then interpreted
// one by one as usual.  Records and report items are emitted exactly as the per-node path does.
bool try_tail(Exec &x)
{
    mg_cycle_plan *p = x.p;
    LevelList &cycle = *p->levels;
    if (p->con_N != 1) return false;
    LevelNode *top = cycle.last();
    // A batched schedule starts its coarse tail one level further down: a W-cycle's 64 slices from N = 64 are 64 workgroups
    // of 130 us (8 solves, 28 smoothing nodes each), its 128 slices from N = 32 are half as long on twice as many CUs, and the
    // 128 visits of level 64 are two batched launches of the register-tile kernel (measured, W(3,3) at 8192^2: tail from
    // 64 / 32 / 16: 1.088 / 1.045 / 1.053 ms)
    static const int batch_tail_n = [] { const char *e = getenv("MG_BATCH_TAIL_N"); return e ? atoi(e) : 32; }();
    if (x.tracing && top->N > batch_tail_n) return false;

    k::TailArgs a;
    int node_level[k::TAIL_MAX_NODES];
    size_t tok = x.tok;
    if (!scan_tail(p->tokens, &tok, p->sizes, x.at, p->con_step, top->N, p->L, &a, node_level)) return false;
    const int n_nodes = a.n_nodes;
    if (p->records.size() + (size_t)n_nodes > p->err_cap) return false;  // (cannot happen: err_cap counts every node token)
    // the slice's records are consecutive: its error slots count from the first of them (so that every instance of a
    // batched launch can run the same node program on an error array of its own)
    const int rec0 = (int)p->records.size();
    // records + report, in the order the per-node interpreter would emit them
    const char *down = "             *\n             |\n Restriction |\n             |\n             *\n";
    const char *up = "             *\n             |\nProlongation |\n             |\n             *\n";
    for (int i = 0; i < n_nodes; ++i) {
        k::TailNode &nd = a.nodes[i];
        const int N = a.N[node_level[i]];
        if (nd.type == -1) {
            const int rec = add_record(p, -1, N, nd.steps);
            nd.err_slot = rec - rec0;
            report_smoothing(p, rec);
            report_text(p, down);
        } else if (nd.type == 0) {
            add_record(p, 0, N, 0);
            if (p->flags & MG_CYCLE_REPORT) {
                Text t;
                t.printf("          ~Exact Solver~\n");
                t.printf("Current Grid Size N = %d\n", N);
                t.printf("   Use Exact Solver = GaussSeidel Even / Odd\n");
                t.printf("       Target Error = %.3e\n", nd.tol);
                report_text(p, t.s.c_str());
            }
        } else {
            const int rec = add_record(p, 1, N, nd.steps);
            nd.err_slot = rec - rec0;
            report_text(p, up);
            report_smoothing(p, rec);
        }
    }
    if (p->flags & MG_CYCLE_MIXED) {
        k::TailArgsF f = tail_args_f32(a);
        f.F_top = (const float *)top->F;
        f.U_top = (float *)top->U;
        f.err_dev = p->err_dev + rec0;
        f.gs_state = x.c.gs_state;
        ProfScope ps("coarse_tail_f32", top->N, 0.0);
        k::tail_launch_f32(x.c.stream, f);
        x.tok = tok;
        return true;
    }
    a.F_top = top->F;
    a.U_top = top->U;
    a.err_dev = p->err_dev + rec0;
    a.gs_state = x.c.gs_state;
    if (x.tracing) {  // described, not launched (its exact-solver state slot is assigned when the schedule is built)
        NodeOp op;
        op.kind = 2;
        op.N = top->N;
        op.F = top->F;
        op.dst = top->U;
        op.tail = (int)p->tails.size();
        p->tails.push_back(a);
        p->ops.push_back(op);
        x.tok = tok;
        return true;
    }
    // diagnostics: MG_TAIL_TRACE=1 prints the in-kernel timeline of the first traced launches
    static const bool trace_on = getenv("MG_TAIL_TRACE") != nullptr;
    static int traced = 0;
    long long *trace = nullptr;
    if (trace_on && traced < 3 && !x.capturing) {
        (void)hipMalloc((void **)&trace, (k::TAIL_MAX_NODES + 2 + 24 * 8) * sizeof(long long));
        (void)hipMemset(trace, 0, (k::TAIL_MAX_NODES + 2 + 24 * 8) * sizeof(long long));
        a.trace = trace;
    }
    {
        ProfScope ps("coarse_tail", top->N, 0.0);
        k::tail_launch(x.c.stream, a);
    }
    if (trace) {
        std::vector<long long> t((size_t)a.n_nodes + 2);
        (void)hipStreamSynchronize(x.c.stream);
        (void)hipMemcpy(t.data(), trace, t.size() * sizeof(long long), hipMemcpyDeviceToHost);
        ++traced;
        fprintf(stderr, "[tail trace] staging %.2f us;", (double)(t[1] - t[0]) * 0.01);
        for (int i = 0; i < a.n_nodes; ++i)
            fprintf(stderr, " %d@%d:%.2f", a.nodes[i].type, a.N[node_level[i]], (double)(t[(size_t)i + 2] - t[(size_t)i + 1]) * 0.01);
        fprintf(stderr, " us\n");
#ifdef MG_TAIL_PHASES   // shader-clock cycles between the stamps of thread 0 inside the first nodes (mg_tail_impl.h: PHASE)
        std::vector<long long> ph(24 * 8);
        (void)hipMemcpy(ph.data(), trace + k::TAIL_MAX_NODES + 2, ph.size() * sizeof(long long), hipMemcpyDeviceToHost);
        for (int i = 0; i < a.n_nodes && i < 24; ++i) {
            if (a.nodes[i].type == 0) continue;
            fprintf(stderr, "[tail phases] %2d@%-3d cycles:", a.nodes[i].type, a.N[node_level[i]]);
            for (int k = 1; k < 8 && ph[(size_t)i * 8 + k]; ++k) fprintf(stderr, " %lld", ph[(size_t)i * 8 + k] - ph[(size_t)i * 8 + k - 1]);
            fprintf(stderr, "\n");
        }
#endif
        (void)hipFree(trace);
    }
    x.tok = tok;
    return true;
}

// one pass over the node program == the reference's while loop :158-426
void run_nodes(Exec &x)
{
    mg_cycle_plan *p = x.p;
    LevelList &cycle = *p->levels;
    const bool fused = (p->flags & MG_CYCLE_FUSED) != 0;
    const bool mixed = (p->flags & MG_CYCLE_MIXED) != 0;
    x.c.defer_norms = true;
    // what a node must be for the dataflow trace: ONE fused launch (smooth_pp records it) -- checked here wherever the
    // entry point below would otherwise fall back to operator-by-operator launches
    auto fusable_down = [&](int N, int M, int step) {
        const RestrictTable &rt = restrict_table(N, M);
        return step >= 1 && step <= k::stream_max_steps() && x.c.smoother != SMOOTHER_SIMPLE && k::stream_fusable(N) && rt.lo && rt.fusable;
    };
    auto fusable_up = [&](int Nc, int N, int step) {
        const ProlongTable &pt = prolong_table(Nc, N);
        return step >= 1 && step <= k::stream_max_steps() && x.c.smoother != SMOOTHER_SIMPLE && k::stream_fusable(N) && pt.owner_row && pt.fusable;
    };

    for (;;) {
        int node;
        if (!x.next_int(&node)) break;  // end of file without a 2
        if (node == 2) break;           // :162-164
        if (x.c.last_error) { x.status = 10; break; }
        if (p->records.size() >= p->err_cap) { x.status = 11; break; }  // error slots exhausted (as the slab driver)

        if (node == -1) {  // :169-300
            int step = 0, next_N = 0;
            if (p->con_step == 0) { if (!x.next_int(&step)) { x.status = 3; break; } }
            else step = p->con_step;
            if (p->con_N == 0) { if (!x.next_int(&next_N)) { x.status = 3; break; } }
            else {
                if (x.at + 1 >= (int)p->sizes.size()) { x.status = 4; break; }  // reference: out-of-bounds read (D2)
                next_N = p->sizes[++x.at];
            }
            if (step == 0) continue;  // :241-243, :296-299 (FMG stub)
            if (next_N < 3) { x.status = 7; break; }
            if (x.tracing && (step < 1 || !fused || mixed)) { x.untraceable = true; break; }

            LevelNode *lv = cycle.last();
            const bool keep = (cycle.Get_init() == 0 && cycle.Is_firstNode());  // :209-214, :252-257
            int rec;
            if (step == -1) {
                if (x.capturing) { x.status = 8; break; }
                if (!keep) mg_fill_zero(lv->U, (size_t)lv->N * lv->N);
                const int done = trigger_smoothing(x, lv);
                rec = add_record(p, -1, lv->N, done);
                p->records[rec].error = lv->smoothingError;
                mg_upload(error_slot(p, rec), &lv->smoothingError, 1);
                mg_getResidual(lv->N, p->L, lv->U, lv->F, lv->D);  // :239
                mg_negate(lv->N, lv->D);                           // :277-280
                report_smoothing(p, rec);
                cycle.Push_back(next_N);  // :283
                if (!cycle.last()->U || !cycle.last()->F || !cycle.last()->D) { x.status = 14; break; }  // out of device memory
                mg_restrict_signed(lv->N, lv->D, next_N, cycle.last()->F, +1);  // :287
            } else if (mixed) {
                // the same fused node with fp32 fields (zero start only: no restart in this mode)
                if (keep) { x.status = 15; break; }
                rec = add_record(p, -1, lv->N, step);
                report_smoothing(p, rec);
                cycle.Push_back(next_N);  // :283
                if (!cycle.last()->U || !cycle.last()->F || !cycle.last()->D) { x.status = 14; break; }
                lv->pending_pre = 0;
                if (p->con_step > 0 && recompute_available(next_N, lv->N, step, p->con_step)) {
                    // (as in the fp64 driver below: the level's `1` node redoes these sweeps, U is neither written nor read)
                    smooth_restrict_f32_no_out(lv->N, p->L, (float *)lv->U, (float *)lv->F, step, error_slot(p, rec), next_N,
                                               (float *)cycle.last()->F);
                    lv->pending_pre = step;
                } else {
                    mg_smooth_restrict_f32(lv->N, p->L, nullptr, (float *)lv->U, (float *)lv->F, step, error_slot(p, rec), next_N,
                                           (float *)cycle.last()->F);
                }
                if (x.c.last_error) { x.status = 15; break; }
                report_text(p, "             *\n             |\n Restriction |\n             |\n             *\n");
                if (!try_tail(x) && next_N <= k::TAIL_MAX_N) { x.status = 15; break; }
                continue;
            } else if (fused) {
                // smoothing (:259), residual (:268), sign flip (:277-280) and restriction (:287)
                // in one pass; the zero fill of U (:256) is folded into the first sweep
                if (x.tracing) {
                    if (keep || !fusable_down(lv->N, next_N, step)) { x.untraceable = true; break; }
                    if (lv->descents > 0) {
                        // a second descent from this level reads its F alone (zero start): an independent task of the
                        // dataflow -- on arrays of its own, so that it can run beside the first one
                        const size_t bytes = (size_t)lv->N * lv->N * sizeof(double);
                        p->pool.put(lv->U);
                        p->pool.put(lv->D);
                        lv->U = (double *)p->pool.get(bytes);
                        lv->D = (double *)p->pool.get(bytes);
                        lv->pending_pre = 0;
                    }
                }
                lv->descents++;
                rec = add_record(p, -1, lv->N, step);
                report_smoothing(p, rec);
                cycle.Push_back(next_N);  // :283
                if (!cycle.last()->U || !cycle.last()->F || !cycle.last()->D || !lv->U || !lv->D) { x.status = 14; break; }  // out of device memory
                double *Fc = cycle.last()->F;
                lv->pending_pre = 0;
                const int up_step = keep ? -1 : matching_up_step(x);
                if (up_step > 0 && recompute_available(next_N, lv->N, step, up_step)) {
                    // the `1` node of this level will redo these sweeps in its own pipeline: U is neither written now
                    // nor read then
                    smooth_restrict_no_out(lv->N, p->L, lv->U, lv->F, step, error_slot(p, rec), next_N, Fc);
                    lv->pending_pre = step;
                } else if (!keep) {
                    mg_smooth_restrict(lv->N, p->L, nullptr, lv->U, lv->F, step, error_slot(p, rec), next_N, Fc);
                } else {
                    double *tmp = (double *)p->pool.get((size_t)lv->N * lv->N * sizeof(double));
                    mg_smooth_restrict(lv->N, p->L, lv->U, tmp, lv->F, step, error_slot(p, rec), next_N, Fc);
                    std::swap(lv->U, tmp);
                    p->pool.put(tmp);
                }
                report_text(p, "             *\n             |\n Restriction |\n             |\n             *\n");
                try_tail(x);  // levels N <= 64: the rest of this descent and its way back up in one launch
                continue;
            } else {
                rec = add_record(p, -1, lv->N, step);
                smooth_level(x, lv, step, !keep, true, rec);
                report_smoothing(p, rec);
                cycle.Push_back(next_N);  // :283
                if (!cycle.last()->U || !cycle.last()->F || !cycle.last()->D) { x.status = 14; break; }  // out of device memory
                mg_restrict_signed(lv->N, lv->D, next_N, cycle.last()->F, +1);  // :287
            }
            report_text(p, "             *\n             |\n Restriction |\n             |\n             *\n");
        } else if (node == 0) {  // :305-324
            double tol;
            int option;
            if (!x.next(&tol) || !x.next_int(&option)) { x.status = 3; break; }
            LevelNode *lv = cycle.last();
            if (x.tracing) { x.untraceable = true; break; }   // a stand-alone exact solve (those of a coarse tail never get here)
            if (mixed) { x.status = 15; break; }  // fp32 coarse solves exist inside the tail kernel only
            if (x.capturing && lv->N > k::gs_single_workgroup_max_n()) { x.status = 8; break; }
            mg_doExactSolver(lv->N, p->L, lv->U, lv->F, tol, option);
            if (x.c.last_error) { x.status = 5; break; }
            add_record(p, 0, lv->N, 0);
            if (p->flags & MG_CYCLE_REPORT) {
                Text t;
                t.printf("          ~Exact Solver~\n");
                t.printf("Current Grid Size N = %d\n", lv->N);
                if (option == 0) t.printf("   Use Exact Solver = Inverse Matrix\n");
                if (option == 1) t.printf("   Use Exact Solver = GaussSeidel Even / Odd\n");
                t.printf("       Target Error = %.3e\n", tol);
                report_text(p, t.s.c_str());
            }
        } else if (node == 1) {  // :329-424
            int step;
            if (p->con_step == 0) { if (!x.next_int(&step)) { x.status = 3; break; } }
            else step = p->con_step;
            if (p->con_N != 0) x.at--;
            if (cycle.depth() < 2) { x.status = 6; break; }

            LevelNode *coarse = cycle.last();
            LevelNode *fine = coarse->prevNode;
            const char *arrow = "             *\n             |\nProlongation |\n             |\n             *\n";
            if (mixed) {
                if (step <= 0) { x.status = 15; break; }
                const int rec = add_record(p, 1, fine->N, step);
                // the node that ends the file on the finest level stores its result in fp64 straight away
                // (the first cycle of a window: a correction cycle's result is ADDED to the iterate, in a pass of its own)
                const bool last_node = fine->N == p->N_max && p->refine_it == 0 && k::stream_fusable(fine->N) &&
                                       prolong_table(coarse->N, fine->N).fusable &&
                                       (x.tok >= p->tokens.size() || (int)p->tokens[x.tok] == 2);
                const int pre = fine->pending_pre;  // > 0: the `-1` node of this level left U to be recomputed here
                if (pre > 0 && !recompute_available(coarse->N, fine->N, pre, step)) { x.status = 15; break; }  // (cannot happen: same test as there)
                fine->pending_pre = 0;
                if (last_node) {
                    if (pre > 0)
                        prolong_smooth_f32_recompute(coarse->N, (const float *)coarse->U, fine->N, p->L, nullptr, p->U64, (const float *)fine->F,
                                                     pre, step, error_slot(p, rec));
                    else
                        prolong_smooth_f32_wide(coarse->N, (const float *)coarse->U, fine->N, p->L, (const float *)fine->U, p->U64,
                                                (const float *)fine->F, step, error_slot(p, rec));
                    x.widened = true;
                } else if (pre > 0) {
                    prolong_smooth_f32_recompute(coarse->N, (const float *)coarse->U, fine->N, p->L, (float *)fine->U, nullptr,
                                                 (const float *)fine->F, pre, step, error_slot(p, rec));
                } else {
                    mg_prolong_smooth_f32(coarse->N, (const float *)coarse->U, fine->N, p->L, (const float *)fine->U, (float *)fine->D,
                                          (float *)fine->F, step, error_slot(p, rec));
                    std::swap(fine->U, fine->D);
                }
                if (x.c.last_error) { x.status = 15; break; }
                cycle.Remove_back();  // :363
                report_text(p, arrow);
                report_smoothing(p, rec);
                continue;
            }
            if (x.tracing && (coarse->pending_pre > 0 || !fused || !fusable_up(coarse->N, fine->N, step) ||
                              (fine->pending_pre > 0 && !recompute_available(coarse->N, fine->N, fine->pending_pre, step)))) {
                x.untraceable = true;
                break;
            }
            ensure_U(p, coarse);
            if (fused && step > 0 && fine->pending_pre > 0 && recompute_available(coarse->N, fine->N, fine->pending_pre, step)) {
                const int rec = add_record(p, 1, fine->N, step);
                prolong_smooth_recompute(coarse->N, coarse->U, fine->N, p->L, fine->U, fine->F, fine->pending_pre, step, error_slot(p, rec));
                fine->pending_pre = 0;
                cycle.Remove_back();  // :363
                report_text(p, arrow);
                report_smoothing(p, rec);
                continue;
            }
            ensure_U(p, fine);
            if (fused && step > 0) {
                // tempU (:353), doProlongation (:354), doGridAddition (:368) and the post-smoothing
                // (:416) in one pass; the fine level's D is dead here and receives the result
                const int rec = add_record(p, 1, fine->N, step);
                mg_prolong_smooth(coarse->N, coarse->U, fine->N, p->L, fine->U, fine->D, fine->F, step,
                                  error_slot(p, rec));
                std::swap(fine->U, fine->D);
                cycle.Remove_back();  // :363
                report_text(p, arrow);
                report_smoothing(p, rec);
                continue;
            }
            if (fused) {
                mg_prolongAdd(coarse->N, coarse->U, fine->N, fine->U, fine->D);
                std::swap(fine->U, fine->D);
                cycle.Remove_back();  // :363
            } else {
                const size_t nf = (size_t)fine->N * fine->N;
                double *tempU = (double *)p->pool.get(nf * sizeof(double));  // :353
                mg_doProlongation(coarse->N, coarse->U, fine->N, tempU);      // :354
                cycle.Remove_back();                                          // :363
                mg_doGridAddition(fine->N, fine->U, tempU);                   // :368
                p->pool.put(tempU);                                           // :371
            }
            report_text(p, arrow);

            if (step == 0) continue;  // :409-411
            LevelNode *lv = cycle.last();
            int rec;
            if (step == -1) {
                if (x.capturing) { x.status = 8; break; }
                const int done = trigger_smoothing(x, lv);
                rec = add_record(p, 1, lv->N, done);
                p->records[rec].error = lv->smoothingError;
                mg_upload(error_slot(p, rec), &lv->smoothingError, 1);
            } else {
                rec = add_record(p, 1, lv->N, step);
                smooth_level(x, lv, step, false, false, rec);
            }
            report_smoothing(p, rec);
        }
        // any other token: ignored, as the reference does
    }
    // Should the level the file ends on still owe its U (pending_pre; today every `-1` node pushes the next level, so
    // the last level never does): materialise it HERE, as part of the node program, so that a captured graph replays it
    // too and the work lies inside the timed window of every run.  fp32 fields have no such launch: an explicit status.
    if (x.status == 0 && cycle.last() && cycle.last()->pending_pre > 0) {
        if (x.tracing) x.untraceable = true;
        else if (mixed) x.status = 15;
        else ensure_U(p, cycle.last());
    }
    flush_norms();  // the window's smoothing errors: one reduction launch
    x.c.defer_norms = false;
}

// reset the level stack to the state right after getSource (:149-153)
void reset_levels(mg_cycle_plan *p)
{
    LevelList &cycle = *p->levels;
    // keep the finest level (its F never changes); drop anything a broken file left over
    while (cycle.depth() > 1) cycle.Remove_back();
    cycle.Set_init(1);
    if (cycle.first()) cycle.first()->descents = 0;
    p->records.clear();
    p->report_items.clear();
}

// ---------------------------------------------------------------------------
// The batched breadth-first schedule (mg_cycle_plan::sched)
// ---------------------------------------------------------------------------
bool same_shape(const mg_cycle_plan *p, const NodeOp &a, const NodeOp &b)
{
    if (a.kind != b.kind || a.N != b.N) return false;
    if (a.kind == 2) {
        const k::TailArgs &x = p->tails[(size_t)a.tail], &y = p->tails[(size_t)b.tail];
        if (x.n_levels != y.n_levels || x.n_nodes != y.n_nodes) return false;
        for (int l = 0; l < x.n_levels; ++l)
            if (x.N[l] != y.N[l]) return false;
        for (int i = 0; i < x.n_nodes; ++i) {
            const k::TailNode &m = x.nodes[i], &n = y.nodes[i];
            if (m.type != n.type || m.steps != n.steps || m.err_slot != n.err_slot || m.tol != n.tol) return false;
        }
        return true;
    }
    return a.take == b.take && a.pre == b.pre && a.no_out == b.no_out && a.d_sign == b.d_sign && (a.src == nullptr) == (b.src == nullptr) &&
           (a.coarse == nullptr) == (b.coarse == nullptr) && (a.Fc == nullptr) == (b.Fc == nullptr) && a.Nc == b.Nc && a.M == b.M && a.L == b.L;
}

void drop_schedule(mg_cycle_plan *p)
{
    p->sched.clear();
    p->ops.clear();
    p->tails.clear();
    p->sched_ready = false;
    if (p->sched_items) (void)hipFree(p->sched_items);
    p->sched_items = nullptr;
    if (p->gs_slots) p->pool.put(p->gs_slots);
    p->gs_slots = nullptr;
    p->last_tail = -1;
    p->pool.park(false);
    p->pool.release_parked();
}

// Trace the file once, build the dataflow depths, merge equal nodes of equal depth.  On success the plan keeps every array
// the trace handed out (one set per visit: 6 n_0 doubles for a W-cycle, 3 GiB at 8192^2) and mg_cycle_enqueue replays
// p->sched; otherwise everything is returned and the interpreter runs the file node by node as before.
void build_schedule(mg_cycle_plan *p)
{
    static const bool on = [] { const char *e = getenv("MG_CYCLE_BATCH"); return !e || atoi(e) != 0; }();
    p->sched_tried = true;
    const bool mixed = (p->flags & MG_CYCLE_MIXED) != 0, fused = (p->flags & MG_CYCLE_FUSED) != 0;
    if (!on || !fused || mixed || p->con_N != 1 || p->con_step < 1 || getenv("MG_NO_TAIL")) return;
    // worth it only where the file descends again from a level it has come back up to
    {
        std::vector<int> descents(1, 0);
        bool points = false;
        for (size_t tok = 0; tok < p->tokens.size() && !points;) {
            const int node = (int)p->tokens[tok++];
            if (node == 2) break;
            if (node == -1) {
                if (descents.back() > 0) points = true;
                descents.back()++;
                descents.push_back(0);
            } else if (node == 0) {
                tok += 2;
            } else if (node == 1) {
                if (descents.size() < 2) break;
                descents.pop_back();
            }
        }
        if (!points) return;
    }
    Context &c = ctx();
    reset_levels(p);
    c.active_pool = &p->pool;
    p->pool.park(true);   // nothing the trace returns is handed out again: every visit keeps arrays of its own
    c.trace = &p->ops;
    c.trace_failed = false;
    // (measured, W(3,3) at 8192^2: the recomputing pair from 4096 / 2048 / 1024 / 512 on: 1.194 / 1.158 / 1.143 / 1.158 ms)
    static const int batch_recompute_min = [] { const char *e = getenv("MG_BATCH_RECOMPUTE_MIN_N"); return e ? atoi(e) : 1024; }();
    if (!getenv("MG_RECOMPUTE_MIN_N")) c.recompute_min_override = batch_recompute_min;
    Exec x{p, c};
    x.tracing = true;
    run_nodes(x);
    c.recompute_min_override = 0;
    c.trace = nullptr;
    c.active_pool = nullptr;
    const bool ok = x.status == 0 && !x.untraceable && !c.trace_failed && !c.last_error && !p->ops.empty();
    c.trace_failed = false;
    if (!ok) {
        drop_schedule(p);
        return;
    }
    // dataflow depth: behind the last writer of everything the node reads, and behind every earlier reader or writer of
    // what it writes (the trace hands out fresh arrays, so the latter only orders a node behind its own inputs' users)
    std::map<const void *, int> wrote, read;
    auto depth_of = [](const std::map<const void *, int> &m, const void *k) {
        const auto it = m.find(k);
        return it == m.end() ? 0 : it->second;
    };
    std::vector<int> depth(p->ops.size(), 0);
    for (size_t i = 0; i < p->ops.size(); ++i) {
        const NodeOp &o = p->ops[i];
        const void *in[3] = {o.F, (o.kind != 2 && o.pre == 0) ? (const void *)o.src : nullptr, o.coarse};
        const void *out[2] = {(o.kind == 2 || !o.no_out) ? (const void *)o.dst : nullptr, o.Fc};
        int d = 0;
        for (const void *q : in)
            if (q) d = std::max(d, depth_of(wrote, q));
        for (const void *q : out)
            if (q) d = std::max(d, std::max(depth_of(wrote, q), depth_of(read, q)));
        depth[i] = ++d;
        for (const void *q : in)
            if (q) read[q] = std::max(depth_of(read, q), d);
        for (const void *q : out)
            if (q) wrote[q] = d;
    }
    // groups: same depth, same shape (file order inside a group; groups in order of depth, then of their first node)
    for (size_t i = 0; i < p->ops.size(); ++i) {
        mg_cycle_plan::Group *g = nullptr;
        for (auto &cand : p->sched)
            if (cand.depth == depth[i] && same_shape(p, p->ops[(size_t)cand.first], p->ops[i])) { g = &cand; break; }
        if (!g) {
            p->sched.emplace_back();
            g = &p->sched.back();
            g->depth = depth[i];
            g->first = (int)i;
        }
        g->members.push_back((int)i);
    }
    std::stable_sort(p->sched.begin(), p->sched.end(), [](const mg_cycle_plan::Group &a, const mg_cycle_plan::Group &b) {
        return a.depth != b.depth ? a.depth < b.depth : a.first < b.first;
    });
    p->sched_max_batch = 0;
    for (const auto &g : p->sched) p->sched_max_batch = std::max(p->sched_max_batch, (int)g.members.size());
    if (p->sched_max_batch < 2) {   // nothing to merge: the interpreter does the same launches with less memory
        drop_schedule(p);
        return;
    }
    // exact-solver state: a slot per coarse-tail instance; the instance tables of all groups in one device array
    if (!p->tails.empty()) {
        p->gs_slots = (int *)p->pool.get(2 * p->tails.size() * sizeof(int));
        if (!p->gs_slots) { drop_schedule(p); return; }
        for (size_t t = 0; t < p->tails.size(); ++t) p->tails[t].gs_state = p->gs_slots + 2 * t;
    }
    std::vector<char> host;
    for (auto &g : p->sched) {
        g.items_at = host.size();
        for (int m : g.members) {
            const NodeOp &o = p->ops[(size_t)m];
            if (o.kind == 2) {
                const k::TailArgs &a = p->tails[(size_t)o.tail];
                const TailBatchItem it{a.F_top, a.U_top, a.err_dev, a.gs_state};
                host.insert(host.end(), (const char *)&it, (const char *)&it + sizeof it);
                p->last_tail = std::max(p->last_tail, o.tail);
            } else {
                const NodeBatchItem it{o.src, o.F, o.coarse, o.dst, o.Fc};
                host.insert(host.end(), (const char *)&it, (const char *)&it + sizeof it);
                g.errs.push_back(o.err);
            }
        }
        while (host.size() % 16) host.push_back(0);
    }
    if (hipMalloc(&p->sched_items, host.size()) != hipSuccess || hipMemcpy(p->sched_items, host.data(), host.size(), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipGetLastError();
        drop_schedule(p);
        return;
    }
    p->pool.park(false);   // (what the trace set aside stays set aside: those are the arrays of the schedule)
    p->sched_U = p->final_U = p->levels->last()->U;
    p->sched_N = p->final_N = p->levels->last()->N;
    p->sched_smoother = (int)c.smoother;
    p->sched_ready = true;
    if (getenv("MG_CYCLE_DEBUG")) {
        fprintf(stderr, "[cycle] batched schedule: %zu nodes in %zu launches (largest batch %d), %.1f MiB of arrays\n", p->ops.size(), p->sched.size(),
                p->sched_max_batch, (double)p->pool.bytes_held() / 1048576.0);
        for (const auto &g : p->sched)
            fprintf(stderr, "   depth %2d  %-44s N=%5d x%zu\n", g.depth, p->ops[(size_t)g.first].kind == 2 ? "coarse_tail" : p->ops[(size_t)g.first].name,
                    p->ops[(size_t)g.first].N, g.members.size());
    }
}

// one window of a plan with a schedule: its launches, in dataflow order, on the engine's stream
void replay_schedule(mg_cycle_plan *p)
{
    Context &c = ctx();
    c.defer_norms = true;
    // the norms of the window are reduced at its end: a reduction in mid-window recycles the arena of partial sums, and a
    // batch registers its instances one by one (the arena is sized for a whole window when the previous one ends)
    c.norms_at_window_end = true;
    for (const auto &g : p->sched) {
        const NodeOp &o = p->ops[(size_t)g.first];
        const int n = (int)g.members.size();
        if (o.kind == 2) {
            ProfScope ps(n > 1 ? "coarse_tail (batch)" : "coarse_tail", o.N, 0.0);
            k::tail_launch(c.stream, p->tails[(size_t)o.tail], n, (const TailBatchItem *)((const char *)p->sched_items + g.items_at));
        } else if (n == 1) {
            replay_node(o, nullptr);
        } else {
            NodeBatch b;
            b.n = n;
            b.dev = (const NodeBatchItem *)((const char *)p->sched_items + g.items_at);
            b.err_outs = g.errs.data();
            replay_node(o, &b);
        }
    }
    flush_norms();
    c.defer_norms = false;
    c.norms_at_window_end = false;
    // mg_lastExactSolverIterations(): the state of the LAST coarse-tail slice of the file (each instance has its own slot)
    if (p->last_tail >= 0)
        (void)hipMemcpyAsync(c.gs_state, p->gs_slots + 2 * p->last_tail, 2 * sizeof(int), hipMemcpyDeviceToDevice, c.stream);
}

bool uses_trigger(const mg_cycle_plan *p)
{
    if (p->con_step == -1) return true;
    if (p->con_step != 0) return false;
    for (double t : p->tokens)
        if (t == -1.0) return true;  // conservative: a manual step of -1 may appear
    return false;
}

}  // namespace

extern "C" {

mg_cycle_plan *mg_cycle_load(const char *path, int flags)
{
    if (!require_ready("mg_cycle_load")) return nullptr;
    std::ifstream f(path);
    if (!f.is_open()) {
        fail(MG_ERR_CYCLE_FILE, "Cannot open file %s", path);  // src/MG_solver_CPU.cpp:65-68
        return nullptr;
    }
    mg_cycle_plan *p = new mg_cycle_plan;
    p->flags = flags;
    p->path = path;
    if (!(f >> p->L >> p->min_x >> p->min_y >> p->con_step >> p->con_N >> p->N_max >> p->N_min)) {
        fail(MG_ERR_CYCLE_FILE, "%s: malformed cycle structure header", path);
        delete p;
        return nullptr;
    }
    if (p->N_max < 3 || p->N_max > 46340 || (p->con_N != 0 && p->N_min < 1)) {
        fail(MG_ERR_CYCLE_FILE, "%s: unusable grid sizes N_max=%d N_min=%d", path, p->N_max, p->N_min);
        delete p;
        return nullptr;
    }
    std::string tok;
    while (f >> tok) {
        char *end = nullptr;
        const double v = strtod(tok.c_str(), &end);
        if (end == tok.c_str()) break;  // a non-numeric token ends the program
        p->tokens.push_back(v);
    }
    if (p->con_N == 1) {  // :111-131
        for (int n = p->N_max; n >= p->N_min && n > 0; n /= 2) p->sizes.push_back(n);
    } else if (p->con_N == 2) {  // :132-146
        for (int n = p->N_max; n >= p->N_min; --n) p->sizes.push_back(n);
    }
    // one error slot per record; every -1, 0 and 1 node adds at most one record (operands that happen to
    // equal a node code only over-count), so the slots cannot run out
    size_t nodes = 0;
    for (double t : p->tokens)
        if (t == -1.0 || t == 1.0 || t == 0.0) ++nodes;
    p->err_cap = nodes + 8;
    p->err_dev = (double *)p->pool.get(p->err_cap * sizeof(double));
    const bool mixed = (flags & MG_CYCLE_MIXED) != 0;
    if (mixed && (!(flags & MG_CYCLE_FUSED) || p->con_N != 1 || p->con_step < 1 || p->con_step > k::stream_max_steps())) {
        fail(MG_ERR_UNSUPPORTED, "%s: the mixed-precision mode needs the fused driver, con_N = 1 and a fixed con_step in 1..%d",
             path, k::stream_max_steps());
        delete p;
        return nullptr;
    }
    p->levels = new LevelList(&p->pool, mixed ? sizeof(float) : sizeof(double));
    p->levels->Push_back(p->N_max);  // :149
    LevelNode *top = p->levels->last();
    if (!top->U || !top->F || !top->D || !p->err_dev) {
        mg_cycle_destroy(p);
        return nullptr;
    }
    if (mixed) {
        // F is evaluated in fp64 exactly as always and rounded to fp32 ONCE; the result comes back
        // widened into an fp64 array
        const size_t n = (size_t)top->N * top->N;
        p->F64 = (double *)p->pool.get(n * sizeof(double));
        p->U64 = (double *)p->pool.get(n * sizeof(double));
        if (!p->F64 || !p->U64) {
            mg_cycle_destroy(p);
            return nullptr;
        }
        mg_getSource(top->N, p->L, p->F64, p->min_x, p->min_y);  // :153
        mg_to_f32((float *)top->F, p->F64, n);
    } else {
        mg_getSource(top->N, p->L, top->F, p->min_x, p->min_y);  // :153
    }
    p->F_finest = top->F;
    (void)hipEventCreate(&p->ev0);
    (void)hipEventCreate(&p->ev1);
    mg_sync();
    return p;
}

// enqueue one run of the window on the engine's stream; no host synchronisation (fixed-step
// cycle files).  Several windows may be enqueued back to back; mg_cycle_collect() then waits and
// reports the LAST one.
int mg_cycle_enqueue(mg_cycle_plan *p)
{
    if (!require_ready("mg_cycle_enqueue") || !p) return 1;
    Context &c = ctx();
    hipStream_t s = c.stream;
    int status = 0;
    // first window: does the file's dataflow allow a batched schedule (independent visits of a level in one launch)?
    if (!p->sched_tried) build_schedule(p);
    const bool want_graph = (p->flags & MG_CYCLE_GRAPH) && (p->flags & MG_CYCLE_FUSED) && !(p->flags & MG_CYCLE_MIXED) &&
                            !uses_trigger(p) && !p->graph_failed && (!p->sched_ready || (int)c.smoother == p->sched_smoother);
    c.profile_window++;
    (void)hipEventRecord(p->ev0, s);

    p->last_replayed = false;
    if (want_graph && p->graph_ready) {
        // the node program is static: replay it.  records/report keep their structure,
        // only the error values are refreshed at collect time.
        if (!MG_HIP(hipGraphLaunch(p->graph_exec, s))) status = 9;
        p->last_replayed = status == 0;
    } else {
        const bool mixed = (p->flags & MG_CYCLE_MIXED) != 0;
        const int outer = mixed ? p->refinements : 1;
        const size_t n_top = (size_t)p->N_max * p->N_max;
        if (mixed && p->F32_stale) {  // a previous window left a residual in the fp32 source
            k::convert_to_f32(s, (float *)p->levels->first()->F, p->F64, n_top);
            p->F32_stale = false;
        }
        // (mg_set_smoother after the trace: the schedule holds the trace's kernels, the interpreter follows the setting)
        const bool use_sched = p->sched_ready && (int)c.smoother == p->sched_smoother;
        if (use_sched) {
            // the schedule is static (arrays, records and report were fixed by the trace): its launches, nothing else
            p->final_U = p->sched_U;
            p->final_N = p->sched_N;
            const bool capture_now = want_graph && p->warm_runs >= 1;  // run 0 warms the tables and the norm arena
            bool capturing = false;
            if (capture_now) {
                if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) capturing = true;
                else { (void)hipGetLastError(); p->graph_failed = true; }
            }
            replay_schedule(p);
            status = c.last_error ? 10 : 0;
            if (capturing) {
                hipGraph_t g = nullptr;
                const hipError_t e = hipStreamEndCapture(s, &g);
                if (e == hipSuccess && status == 0 && hipGraphInstantiate(&p->graph_exec, g, nullptr, nullptr, 0) == hipSuccess) {
                    p->graph = g;
                    p->graph_ready = true;
                    if (!MG_HIP(hipGraphLaunch(p->graph_exec, s))) status = 9;
                    p->last_replayed = status == 0;
                } else {
                    (void)hipGetLastError();
                    if (g) (void)hipGraphDestroy(g);
                    p->graph_failed = true;
                    replay_schedule(p);   // an eager pass, so that this window still produces the result
                    status = c.last_error ? 10 : 0;
                }
            }
            p->warm_runs++;
        }
        for (int it = 0; it < outer && status == 0 && !use_sched; ++it) {
        reset_levels(p);  // records/report: those of the last fp32 cycle
        c.active_pool = &p->pool;
        p->refine_it = it;
        if (it > 0) {
            // fp64 residual of the fp64 iterate -> fp32 source of the next correction cycle (in its own array when there
            // is one: the finest level then simply points at it for this cycle)
            const double dx = p->L / (double)(p->N_max - 1);
            float *dst = p->F32_res ? p->F32_res : (float *)p->levels->first()->F;
            k::refine_residual(s, p->N_max, 1.0 / (dx * dx), p->U64, p->F64, dst, p->refine_err_dev + (it - 1));
            if (p->F32_res) p->levels->first()->F = (double *)p->F32_res;
            else p->F32_stale = true;
        }
        Exec x{p, c};
        const bool capture_now = want_graph && p->warm_runs >= 1;  // run 0 warms pool + tables
        if (capture_now) {
            if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) x.capturing = true;
            else { (void)hipGetLastError(); p->graph_failed = true; }
        }
        static const bool dbg = getenv("MG_CYCLE_DEBUG") != nullptr;
        if (dbg) fprintf(stderr, "[cycle] run_nodes: capturing=%d\n", (int)x.capturing);
        run_nodes(x);
        status = x.status;
        if (dbg) fprintf(stderr, "[cycle] run_nodes done: status %d\n", status);
        if (x.capturing) {
            hipGraph_t g = nullptr;
            const hipError_t e = hipStreamEndCapture(s, &g);
            if (dbg) fprintf(stderr, "[cycle] end capture: %s\n", hipGetErrorString(e));
            if (e == hipSuccess && status == 0 && hipGraphInstantiate(&p->graph_exec, g, nullptr, nullptr, 0) == hipSuccess) {
                if (dbg) fprintf(stderr, "[cycle] instantiated\n");
                p->graph = g;
                p->graph_ready = true;
                if (!MG_HIP(hipGraphLaunch(p->graph_exec, s))) status = 9;
                p->last_replayed = status == 0;
                if (dbg) fprintf(stderr, "[cycle] launched: status %d\n", status);
            } else {
                (void)hipGetLastError();
                if (g) (void)hipGraphDestroy(g);
                p->graph_failed = true;
                if (status == 8) status = 0;
                // fall back to an eager pass so this window still produces the result
                reset_levels(p);
                Exec y{p, c};
                run_nodes(y);
                status = y.status;
            }
        }
        c.active_pool = nullptr;
        p->warm_runs++;
        p->levels->first()->F = p->F_finest;   // (a correction cycle ran on the residual array)
        LevelNode *last = p->levels->last();
        p->final_U = last->U;
        p->final_N = last->N;
        if (mixed && status == 0 && last->N == p->N_max) {
            // fp64 correction: U64 = (double)e on the first cycle, U64 += (double)e afterwards
            if (x.widened) { /* the last node stored fp64 already */ }
            else if (it == 0) k::convert_to_f64(s, p->U64, (const float *)last->U, n_top);
            else k::add_widened(s, p->U64, (const float *)last->U, n_top);
            p->final_U = p->U64;
        } else if (mixed && status == 0) {
            status = 15;  // the file does not come back to the finest level: nothing to widen
        }
        }
    }
    (void)hipEventRecord(p->ev1, s);
    p->last_status = status;
    return status;
}

// wait for the enqueued windows and report the last one
int mg_cycle_collect(mg_cycle_plan *p, mg_cycle_result *out)
{
    if (!require_ready("mg_cycle_collect") || !p) return 1;
    Context &c = ctx();
    memset(out, 0, sizeof *out);
    mg_sync();
    float dev_ms = 0.f;
    if (hipEventElapsedTime(&dev_ms, p->ev0, p->ev1) != hipSuccess) (void)hipGetLastError();

    // smoothing errors: one download for the whole window
    if (!p->records.empty()) {
        std::vector<double> all(p->records.size());
        mg_download(all.data(), p->err_dev, all.size());
        for (size_t i = 0; i < p->records.size(); ++i)
            if (p->records[i].node != 0) p->records[i].error = all[i];
    }

    p->refine_err.clear();
    if ((p->flags & MG_CYCLE_MIXED) && p->refinements > 1 && p->refine_err_dev) {
        p->refine_err.resize((size_t)p->refinements - 1);
        mg_download(p->refine_err.data(), p->refine_err_dev, p->refine_err.size());
    }

    double mg_error = 0.0;  // outside the reference's timed window as well (:434-445)
    if (p->flags & (MG_CYCLE_ERROR | MG_CYCLE_REPORT))
        mg_analyticError(p->final_N, p->L, p->final_U, p->min_x, p->min_y, &mg_error);

    if (p->flags & MG_CYCLE_REPORT) {
        Text t;
        for (const ReportItem &it : p->report_items) {
            if (it.record < 0) {
                t.s += it.text;
                continue;
            }
            const mg_node_record &r = p->records[it.record];  // :261-264, :418-421
            t.printf("          ~Smoothing~\n");
            t.printf("Current Grid Size N = %d\n", r.N);
            t.printf("    Smoothing Steps = %d\n", r.steps);
            t.printf("              Error = %lf\n", r.error);
        }
        t.printf("\n\n===== Final Result =====\n    Error = %lf\n", mg_error);  // :448-450
        p->report = t.s;
    }

    out->status = p->last_status ? p->last_status : (c.last_error ? 10 : 0);
    out->N = p->final_N;
    out->U_dev = p->final_U;
    out->mg_error = mg_error;
    out->time_ms = 0.0;
    out->device_ms = dev_ms;
    out->n_records = (int)p->records.size();
    out->records = p->records.data();
    out->report = p->report.c_str();
    out->graph_replayed = p->last_replayed ? 1 : 0;
    out->schedule_launches = (p->sched_ready && (int)c.smoother == p->sched_smoother) ? (int)p->sched.size() : 0;
    return out->status;
}

// one synchronous run of the reference's timed window (:156 ... :429)
int mg_cycle_execute(mg_cycle_plan *p, mg_cycle_result *out)
{
    if (!require_ready("mg_cycle_execute") || !p) return 1;
    mg_sync();
    const auto t0 = std::chrono::steady_clock::now();  // :156
    mg_cycle_enqueue(p);
    mg_sync();
    const auto t1 = std::chrono::steady_clock::now();  // :429
    const int status = mg_cycle_collect(p, out);
    out->time_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    return status;
}

// mixed-precision mode: `cycles` fp32 runs of the cycle file per window, joined by the fp64
// residual of the fp64 iterate and an fp64 correction (BASELINE.json configs[4]: "mixed fp32
// smoothing / fp64 residual correction").  cycles = 1 is the plain fp32 cycle.
int mg_cycle_set_refinement(mg_cycle_plan *p, int cycles)
{
    if (!require_ready("mg_cycle_set_refinement") || !p) return 1;
    if (!(p->flags & MG_CYCLE_MIXED) || cycles < 1 || cycles > 64) {
        fail(MG_ERR_ARG, "mg_cycle_set_refinement: needs a MG_CYCLE_MIXED plan and 1..64 cycles (got %d)", cycles);
        return 1;
    }
    if (!p->refine_err_dev) {
        p->refine_err_dev = (double *)p->pool.get(64 * sizeof(double));
        if (!p->refine_err_dev) return 1;
    }
    if (cycles > 1 && !p->F32_res) p->F32_res = (float *)p->pool.get((size_t)p->N_max * p->N_max * sizeof(float));   // (nullptr: the source array is reused and re-made)
    p->refinements = cycles;
    return 0;
}

// doSmoothing's error metric (:607-622) of the fp64 iterate after 1, 2, ... cycles-1 corrections
// (evaluated where the refinement forms its residual); returns how many were written
int mg_cycle_refinement_errors(mg_cycle_plan *p, double *out, int cap)
{
    if (!p || !out) return 0;
    int n = 0;
    for (; n < (int)p->refine_err.size() && n < cap; ++n) out[n] = p->refine_err[(size_t)n];
    return n;
}

void mg_cycle_destroy(mg_cycle_plan *p)
{
    if (!p) return;
    if (ctx().ready) (void)hipStreamSynchronize(ctx().stream);
    if (p->graph_exec) (void)hipGraphExecDestroy(p->graph_exec);
    if (p->graph) (void)hipGraphDestroy(p->graph);
    if (p->ev0) (void)hipEventDestroy(p->ev0);
    if (p->ev1) (void)hipEventDestroy(p->ev1);
    if (p->levels) {
        p->levels->clear();
        delete p->levels;
    }
    drop_schedule(p);   // (the arrays the trace set aside go back to the pool)
    if (p->err_dev) p->pool.put(p->err_dev);
    if (p->refine_err_dev) p->pool.put(p->refine_err_dev);
    if (p->F32_res) p->pool.put(p->F32_res);
    if (p->F64) p->pool.put(p->F64);
    if (p->U64) p->pool.put(p->U64);
    p->pool.trim();
    delete p;
}

// src/MG_solver_CPU.cpp:735-754 on a device array: top row first, "%lf" comma separated
int mg_print2File(int N, const double *U_dev, const char *file_name)
{
    if (!require_ready("mg_print2File")) return 1;
    std::vector<double> U((size_t)N * N);
    mg_download(U.data(), U_dev, U.size());
    FILE *o = fopen(file_name, "w");
    if (!o) {
        fail(MG_ERR_ARG, "Cannot open file %s", file_name);
        return 1;
    }
    for (int j = N - 1; j >= 0; --j)
        for (int i = 0; i < N; ++i) fprintf(o, i == N - 1 ? "%lf\n" : "%lf,", U[i + (size_t)N * j]);
    fclose(o);
    return 0;
}

// The reference program (src/MG_solver_CPU.cpp:36-462) on the engine:
//   MG_HIP N_THREADS_OMP cycle_filename.txt
// argv[1] is accepted for command-line compatibility (there are no OpenMP loops left).
int mg_cycle_main(int argc, char **argv)
{
    if (argc != 3) {  // :51-54
        printf("[ ERROR ]: Wrong input numbers of parameter.\n");
        return 1;
    }
    printf("OpenMP threads = %d\n", atoi(argv[1]));       // :59
    printf("Cycle structure file name = %s\n", argv[2]);  // :63
    const char *dev = getenv("MG_DEVICE");
    if (mg_init(dev ? atoi(dev) : 0) != 0) return 1;
    // MG_MIXED=1: the fp32 cycle (what the reference's MG_GPU program computes in);  MG_REFINE=k: k such
    // cycles joined by an fp64 residual and correction.  Same report and CSV.
    const char *mixed = getenv("MG_MIXED"), *refine = getenv("MG_REFINE");
    const bool want_mixed = (mixed && atoi(mixed) != 0) || (refine && atoi(refine) > 1);
    mg_cycle_plan *plan = mg_cycle_load(argv[2], MG_CYCLE_FUSED | MG_CYCLE_REPORT | MG_CYCLE_ERROR | (want_mixed ? MG_CYCLE_MIXED : 0));
    if (!plan) {
        // (a failed call has already printed its reason to stderr and, by default, exited)
        printf("[ ERROR ]: Cannot open file %s\n", argv[2]);  // :66
        return 1;
    }
    if (refine && atoi(refine) > 1 && mg_cycle_set_refinement(plan, atoi(refine)) != 0) {
        printf("[ ERROR ]: %s\n", mg_last_error_string());
        mg_cycle_destroy(plan);
        return 1;
    }
    mg_cycle_result res;
    // MG_WARMUP=k: k untimed runs first.  A process's first window also pays for loading the code objects, the first
    // hipMalloc of every level (the pool is empty) and the upload of the transfer tables: ~10 ms whatever the grid.
    if (const char *w = getenv("MG_WARMUP"))
        for (int i = 0; i < atoi(w); ++i) mg_cycle_execute(plan, &res);
    const int status = mg_cycle_execute(plan, &res);
    if (status != 0) {
        printf("[ ERROR ]: cycle structure file is malformed (status %d)\n", status);
        mg_cycle_destroy(plan);
        return 1;
    }
    fputs(res.report, stdout);
    printf("Time Used = %lf (ms)\n", res.time_ms);  // :451
    // output name: "Sol_" + backend tag + argv[2], as the reference builds "Sol_CPU_"/"Sol_GPU_" (:454-456)
    std::string name = std::string("Sol_HIP_") + argv[2];
    mg_print2File(res.N, res.U_dev, name.c_str());
    printf("Output file name = %s\n", name.c_str());  // :459
    mg_cycle_destroy(plan);
    mg_finalize();
    return 0;
}

}  // extern "C"
