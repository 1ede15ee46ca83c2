// mg_comm.cpp -- RCCL transport for the 1-D row-slab decomposition (one process per GPU,
// BASELINE.json north_star: "two-row ghost exchange via RCCL over xGMI").
//
// The reference has no distributed code at all ("Collective / NCCL call
// sites: none"), so nothing here translates a reference call pattern.  The pattern is sized
// for xGMI point-to-point links: a rank only ever talks to its two slab neighbours
// (ncclSend/ncclRecv grouped, 2 of the 7 links), ghost messages are a few contiguous rows
// (rows are contiguous in the row-major layout), and because the smoother is temporally
// blocked ONE exchange of S+2 ghost rows feeds S sweeps + residual + restriction -- the
// message count per V-cycle level is 2, not one per sweep.
//
// A second, host-staged transport (mg_comm_init_host) carries the SAME calls of the slab driver
// over two callbacks of the embedding program (an MPI or gloo wire): it exists so that the real
// rank-mode driver -- counts, peers, grouping, the collapse all-gather -- can be run by several
// processes on ONE GPU box and compared with the oracle, which RCCL (one rank per device) cannot
// do there.  It moves ghost rows only; all arithmetic stays on the device.
//
// RCCL is NOT a link-time dependency of the engine: its entry points are resolved with dlopen on the
// first communicator call.  A single-GPU user never maps it, and a process that already carries a copy
// (PyTorch bundles its own librccl.so) reuses THAT copy instead of mapping a second RCCL/HIP runtime
// next to it (two runtimes in one process abort at exit).  <rccl/rccl.h> is included for its types only.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <vector>

#include "mg_internal.h"

namespace mg {

namespace {
ncclComm_t g_comm = nullptr;
int g_rank = 0, g_nranks = 1;
hipStream_t g_stream = nullptr;   // comm_set_stream: where the next operations are enqueued (nullptr: the engine's stream)
inline hipStream_t cur_stream() { return g_stream ? g_stream : ctx().stream; }

// ---- RCCL entry points, resolved lazily -------------------------------------------------------
struct Rccl {
    void *handle = nullptr;
    bool tried = false;
    // (pointer types taken from the declarations of <rccl/rccl.h>: a signature cannot drift from the header)
    decltype(&::ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&::ncclCommInitRank) CommInitRank = nullptr;
    decltype(&::ncclCommDestroy) CommDestroy = nullptr;
    decltype(&::ncclGroupStart) GroupStart = nullptr;
    decltype(&::ncclGroupEnd) GroupEnd = nullptr;
    decltype(&::ncclSend) Send = nullptr;
    decltype(&::ncclRecv) Recv = nullptr;
    decltype(&::ncclAllGather) AllGather = nullptr;
    decltype(&::ncclGetErrorString) GetErrorString = nullptr;
    std::string where;
} g_rccl;

// order: MG_RCCL_LIB (explicit path), a copy this process has mapped already (RTLD_NOLOAD matches by
// soname, so torch's bundled "librccl.so" with soname librccl.so.1 is found), the loader's search
// path, the ROCm installation
bool rccl_load()
{
    Rccl &r = g_rccl;
    if (r.handle) return true;
    if (r.tried) return false;
    r.tried = true;
    std::vector<std::pair<std::string, int>> cand;
    if (const char *e = getenv("MG_RCCL_LIB")) cand.emplace_back(e, RTLD_NOW | RTLD_GLOBAL);
    cand.emplace_back("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    cand.emplace_back("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    cand.emplace_back("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (const char *e = getenv("ROCM_PATH")) cand.emplace_back(std::string(e) + "/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    cand.emplace_back("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    for (const auto &c : cand) {
        r.handle = dlopen(c.first.c_str(), c.second);
        if (r.handle) {
            r.where = c.first + ((c.second & RTLD_NOLOAD) ? " (already mapped)" : "");
            break;
        }
    }
    if (!r.handle) {
        fail(MG_ERR_COMM, "RCCL is not available: dlopen(librccl.so.1) failed (%s); set MG_RCCL_LIB", dlerror());
        return false;
    }
    bool ok = true;
    auto sym = [&](const char *name) {
        void *p = dlsym(r.handle, name);
        if (!p) ok = false;
        return p;
    };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    if (!ok) {
        fail(MG_ERR_COMM, "RCCL library %s lacks a required entry point", r.where.c_str());
        dlclose(r.handle);
        r.handle = nullptr;
        return false;
    }
    if (getenv("MG_COMM_DEBUG")) fprintf(stderr, "[mg_comm] RCCL from %s\n", r.where.c_str());
    return true;
}

// host-staged transport
mg_host_transport g_host = {};
bool g_host_on = false;
int g_group_depth = 0;
struct HostOp {
    int is_send, peer;
    void *dev;
    size_t bytes;
};
std::vector<HostOp> g_ops;

// run the queued point-to-point operations of one group: device -> host for the sends, the
// embedding program's exchange (returns when every transfer has completed), host -> device
void host_flush()
{
    if (g_ops.empty()) return;
    (void)hipStreamSynchronize(cur_stream());  // the rows being sent were written by kernels this stream has waited for
    const size_t n = g_ops.size();
    std::vector<std::vector<unsigned char>> stage(n);
    std::vector<int> is_send(n), peer(n);
    std::vector<void *> bufs(n);  // handed over as void *const *
    std::vector<size_t> counts(n);
    for (size_t i = 0; i < n; ++i) {
        const HostOp &o = g_ops[i];
        stage[i].resize(o.bytes);
        if (o.is_send && !MG_HIP(hipMemcpy(stage[i].data(), o.dev, o.bytes, hipMemcpyDeviceToHost))) return;
        is_send[i] = o.is_send;
        peer[i] = o.peer;
        bufs[i] = stage[i].data();
        counts[i] = o.bytes;
    }
    if (g_host.exchange(g_host.user, (int)n, is_send.data(), peer.data(), bufs.data(), counts.data()) != 0) {
        fail(MG_ERR_COMM, "host transport: exchange callback failed");
        g_ops.clear();
        return;
    }
    for (size_t i = 0; i < n; ++i) {
        const HostOp &o = g_ops[i];
        if (!o.is_send && !MG_HIP(hipMemcpy(o.dev, stage[i].data(), o.bytes, hipMemcpyHostToDevice))) break;
    }
    g_ops.clear();
}

bool nccl_ok(ncclResult_t r, const char *what)
{
    if (r == ncclSuccess) return true;
    fail(MG_ERR_COMM, "RCCL call failed: %s -> %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    return false;
}
#define MG_NCCL(expr) nccl_ok((expr), #expr)
}  // namespace

bool comm_ready() { return g_comm != nullptr || g_host_on; }
void comm_set_stream(hipStream_t s) { g_stream = s; }
int comm_rank() { return g_rank; }
int comm_size() { return g_nranks; }

// my rows [send_lo, send_lo+n) go to `peer`; rows from `peer` land at recv.  Either side may
// be empty (n == 0).  All exchanges of one step are issued inside ONE group.
void comm_group_begin()
{
    if (g_host_on) {
        ++g_group_depth;
        return;
    }
    MG_NCCL(g_rccl.GroupStart());
}
void comm_group_end()
{
    if (g_host_on) {
        if (--g_group_depth == 0) host_flush();
        return;
    }
    MG_NCCL(g_rccl.GroupEnd());
}

void comm_send(const void *buf, size_t bytes, int peer)
{
    if (!bytes) return;
    if (g_host_on) {
        g_ops.push_back(HostOp{1, peer, const_cast<void *>(buf), bytes});
        if (g_group_depth == 0) host_flush();
        return;
    }
    MG_NCCL(g_rccl.Send(buf, bytes, ncclChar, peer, g_comm, cur_stream()));
}
void comm_recv(void *buf, size_t bytes, int peer)
{
    if (!bytes) return;
    if (g_host_on) {
        g_ops.push_back(HostOp{0, peer, buf, bytes});
        if (g_group_depth == 0) host_flush();
        return;
    }
    MG_NCCL(g_rccl.Recv(buf, bytes, ncclChar, peer, g_comm, cur_stream()));
}
void comm_allgather(const double *send, double *recv, size_t count_per_rank)
{
    if (g_host_on) {
        (void)hipStreamSynchronize(cur_stream());
        std::vector<double> in(count_per_rank), out(count_per_rank * (size_t)g_nranks);
        if (!MG_HIP(hipMemcpy(in.data(), send, in.size() * sizeof(double), hipMemcpyDeviceToHost))) return;
        if (g_host.allgather(g_host.user, in.data(), out.data(), count_per_rank) != 0) {
            fail(MG_ERR_COMM, "host transport: allgather callback failed");
            return;
        }
        (void)MG_HIP(hipMemcpy(recv, out.data(), out.size() * sizeof(double), hipMemcpyHostToDevice));
        return;
    }
    MG_NCCL(g_rccl.AllGather(send, recv, count_per_rank, ncclDouble, g_comm, cur_stream()));
}

}  // namespace mg

using namespace mg;

extern "C" {

int mg_comm_unique_id_bytes(void) { return (int)sizeof(ncclUniqueId); }

int mg_comm_get_unique_id(void *out)
{
    ncclUniqueId id;
    if (!rccl_load() || !MG_NCCL(g_rccl.GetUniqueId(&id))) return 1;
    memcpy(out, &id, sizeof id);
    return 0;
}

// one communicator per process, on the engine's device and stream
int mg_comm_init(int rank, int nranks, const void *unique_id)
{
    if (!require_ready("mg_comm_init")) return 1;
    if (g_comm) return 0;
    if (nranks < 1 || rank < 0 || rank >= nranks) {
        fail(MG_ERR_ARG, "mg_comm_init: rank %d of %d", rank, nranks);
        return 1;
    }
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    if (!rccl_load() || !MG_NCCL(g_rccl.CommInitRank(&g_comm, nranks, id, rank))) return 1;
    g_rank = rank;
    g_nranks = nranks;
    return 0;
}

// host-staged transport (see the head of this file); the callbacks must stay valid until
// mg_comm_finalize
int mg_comm_init_host(int rank, int nranks, const mg_host_transport *t)
{
    if (!require_ready("mg_comm_init_host")) return 1;
    if (g_comm || g_host_on) return 0;
    if (nranks < 1 || rank < 0 || rank >= nranks || !t || !t->exchange || !t->allgather) {
        fail(MG_ERR_ARG, "mg_comm_init_host: rank %d of %d, or a callback is missing", rank, nranks);
        return 1;
    }
    g_host = *t;
    g_host_on = true;
    g_group_depth = 0;
    g_ops.clear();
    g_rank = rank;
    g_nranks = nranks;
    return 0;
}

void mg_comm_finalize(void)
{
    g_host_on = false;
    g_ops.clear();
    if (g_comm) {
        (void)hipStreamSynchronize(ctx().stream);
        (void)g_rccl.CommDestroy(g_comm);
        g_comm = nullptr;
    }
    g_rank = 0;
    g_nranks = 1;
}

// Exercises the RCCL entry points the slab driver uses -- a grouped send/recv pair (to this rank itself) on a
// second stream ordered by events against the engine's stream, and an all-gather -- and checks the bytes that
// arrived.  Works with a 1-rank communicator, i.e. on a one-GPU box.  Returns 0 when everything matched.
int mg_comm_selftest(size_t n_doubles)
{
    if (!require_ready("mg_comm_selftest")) return 1;
    if (!g_comm) {
        fail(MG_ERR_COMM, "mg_comm_selftest: no RCCL communicator (mg_comm_init)");
        return 1;
    }
    Context &c = ctx();
    const size_t n = n_doubles ? n_doubles : 4096;
    double *a = (double *)c.pool.get(n * sizeof(double)), *b = (double *)c.pool.get(n * sizeof(double));
    double *g = (double *)c.pool.get(n * (size_t)g_nranks * sizeof(double));
    if (!a || !b || !g) return 1;
    k::fill_uniform(c.stream, a, n, 0x5e1f7e57ull + (uint64_t)g_rank);
    (void)hipMemsetAsync(b, 0, n * sizeof(double), c.stream);
    hipStream_t s2 = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = 1;
    if (MG_HIP(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)) && MG_HIP(hipEventCreateWithFlags(&e0, hipEventDisableTiming)) &&
        MG_HIP(hipEventCreateWithFlags(&e1, hipEventDisableTiming))) {
        (void)hipEventRecord(e0, c.stream);
        (void)hipStreamWaitEvent(s2, e0, 0);
        comm_set_stream(s2);
        comm_group_begin();
        comm_send(a, n * sizeof(double), g_rank);
        comm_recv(b, n * sizeof(double), g_rank);
        comm_group_end();
        comm_set_stream(nullptr);
        (void)hipEventRecord(e1, s2);
        (void)hipStreamWaitEvent(c.stream, e1, 0);
        comm_allgather(b, g, n);
        std::vector<double> ha(n), hb(n), hg(n * (size_t)g_nranks);
        mg_download(ha.data(), a, n);
        mg_download(hb.data(), b, n);
        mg_download(hg.data(), g, hg.size());
        rc = (memcmp(ha.data(), hb.data(), n * sizeof(double)) == 0 &&
              memcmp(ha.data(), hg.data() + (size_t)g_rank * n, n * sizeof(double)) == 0 && c.last_error == 0)
                 ? 0 : 2;
    }
    if (s2) (void)hipStreamDestroy(s2);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    c.pool.put(a);
    c.pool.put(b);
    c.pool.put(g);
    return rc;
}

int mg_comm_rank(void) { return g_rank; }
int mg_comm_size(void) { return g_nranks; }

// which wire carries the ghost rows: the file the RCCL entry points were resolved from (dladdr of ncclSend: the absolute
// path the loader mapped, whoever mapped it first), "host transport" for the rehearsal wire, "" before any communicator
const char *mg_comm_library(void)
{
    static std::string text;
    if (g_host_on) return "host transport (mg_comm_init_host)";
    if (!g_rccl.handle) return "";
    Dl_info info;
    memset(&info, 0, sizeof info);
    if (g_rccl.Send && dladdr((void *)g_rccl.Send, &info) && info.dli_fname) text = info.dli_fname;
    else text = g_rccl.where;
    return text.c_str();
}

}  // extern "C"
