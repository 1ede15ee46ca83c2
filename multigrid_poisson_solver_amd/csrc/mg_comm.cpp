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
#include <rccl/rccl.h>

#include <cstring>

#include "mg_internal.h"

namespace mg {

namespace {
ncclComm_t g_comm = nullptr;
int g_rank = 0, g_nranks = 1;

bool nccl_ok(ncclResult_t r, const char *what)
{
    if (r == ncclSuccess) return true;
    fail(MG_ERR_COMM, "RCCL call failed: %s -> %s", what, ncclGetErrorString(r));
    return false;
}
#define MG_NCCL(expr) nccl_ok((expr), #expr)
}  // namespace

bool comm_ready() { return g_comm != nullptr; }
int comm_rank() { return g_rank; }
int comm_size() { return g_nranks; }

// my rows [send_lo, send_lo+n) go to `peer`; rows from `peer` land at recv.  Either side may
// be empty (n == 0).  All exchanges of one step are issued inside ONE group.
void comm_group_begin() { MG_NCCL(ncclGroupStart()); }
void comm_group_end() { MG_NCCL(ncclGroupEnd()); }

void comm_send(const double *buf, size_t count, int peer)
{
    if (count) MG_NCCL(ncclSend(buf, count, ncclDouble, peer, g_comm, ctx().stream));
}
void comm_recv(double *buf, size_t count, int peer)
{
    if (count) MG_NCCL(ncclRecv(buf, count, ncclDouble, peer, g_comm, ctx().stream));
}
void comm_bcast(double *buf, size_t count, int root)
{
    MG_NCCL(ncclBroadcast(buf, buf, count, ncclDouble, root, g_comm, ctx().stream));
}
void comm_allgather(const double *send, double *recv, size_t count_per_rank)
{
    MG_NCCL(ncclAllGather(send, recv, count_per_rank, ncclDouble, g_comm, ctx().stream));
}

}  // namespace mg

using namespace mg;

extern "C" {

int mg_comm_unique_id_bytes(void) { return (int)sizeof(ncclUniqueId); }

int mg_comm_get_unique_id(void *out)
{
    ncclUniqueId id;
    if (!MG_NCCL(ncclGetUniqueId(&id))) return 1;
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
    if (!MG_NCCL(ncclCommInitRank(&g_comm, nranks, id, rank))) return 1;
    g_rank = rank;
    g_nranks = nranks;
    return 0;
}

void mg_comm_finalize(void)
{
    if (g_comm) {
        (void)hipStreamSynchronize(ctx().stream);
        (void)ncclCommDestroy(g_comm);
        g_comm = nullptr;
    }
    g_rank = 0;
    g_nranks = 1;
}

int mg_comm_rank(void) { return g_rank; }
int mg_comm_size(void) { return g_nranks; }

}  // extern "C"
