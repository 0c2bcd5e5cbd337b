// RCCL (xGMI) communicator: one process per GPU.  The only collective on the
// migration path is the all-gather of the trace-major input image before the
// Kirchhoff diffraction sum (outputs are disjoint, no reduction is needed).
#include "common.h"
#include <rccl/rccl.h>

#define IMPDAR_NCCL_CHECK(expr)                                                        \
    do {                                                                               \
        ncclResult_t _r = (expr);                                                      \
        if (_r != ncclSuccess) {                                                       \
            impdar_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,             \
                             ncclGetErrorString(_r));                                  \
            return IMPDAR_ERR_COMM;                                                    \
        }                                                                              \
    } while (0)

static_assert(sizeof(ncclUniqueId) <= IMPDAR_UNIQUE_ID_BYTES, "unique id does not fit the ABI buffer");

extern "C" int impdar_comm_unique_id(char id[IMPDAR_UNIQUE_ID_BYTES])
{
    IMPDAR_ARG_CHECK(id, "null id buffer");
    ncclUniqueId u;
    IMPDAR_NCCL_CHECK(ncclGetUniqueId(&u));
    memset(id, 0, IMPDAR_UNIQUE_ID_BYTES);
    memcpy(id, &u, sizeof(u));
    return IMPDAR_OK;
}

extern "C" int impdar_comm_init(impdar_ctx *ctx, const char id[IMPDAR_UNIQUE_ID_BYTES], int rank, int nranks)
{
    IMPDAR_ARG_CHECK(ctx && id, "null context/id");
    IMPDAR_ARG_CHECK(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank %d of %d", rank, nranks);
    IMPDAR_ARG_CHECK(ctx->comm == nullptr, "communicator already initialised");
    IMPDAR_HIP_CHECK(hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t c = nullptr;
    IMPDAR_NCCL_CHECK(ncclCommInitRank(&c, nranks, u, rank));
    ctx->comm = reinterpret_cast<ncclComm *>(c);
    ctx->rank = rank;
    ctx->nranks = nranks;
    return IMPDAR_OK;
}

void impdar_comm_destroy(impdar_ctx *ctx)
{
    if (ctx && ctx->comm) {
        (void)ncclCommDestroy(reinterpret_cast<ncclComm_t>(ctx->comm));
        ctx->comm = nullptr;
    }
}

extern "C" int impdar_comm_rank(const impdar_ctx *ctx) { return ctx ? ctx->rank : IMPDAR_ERR_ARG; }
extern "C" int impdar_comm_size(const impdar_ctx *ctx) { return ctx ? ctx->nranks : IMPDAR_ERR_ARG; }

// What RCCL itself says about the communicator (not what the caller passed to impdar_comm_init): ranks in it, this
// process's rank, the device it is bound to, and the library version -- so that a bench record of an N-GPU run can
// show that the exchange really ran over an N-rank RCCL communicator.
extern "C" int impdar_comm_info(const impdar_ctx *ctx, int *ranks, int *rank, int *device, int *version)
{
    IMPDAR_ARG_CHECK(ctx && ctx->comm, "communicator not initialised (impdar_comm_init)");
    ncclComm_t c = reinterpret_cast<ncclComm_t>(ctx->comm);
    int v = 0;
    if (ranks) IMPDAR_NCCL_CHECK(ncclCommCount(c, ranks));
    if (rank) IMPDAR_NCCL_CHECK(ncclCommUserRank(c, rank));
    if (device) IMPDAR_NCCL_CHECK(ncclCommCuDevice(c, device));
    if (version) {
        IMPDAR_NCCL_CHECK(ncclGetVersion(&v));
        *version = v;
    }
    return IMPDAR_OK;
}

// In-place all-gather: rank r owns bytes [r*per, (r+1)*per) of `image`.
int impdar_allgather_rows(impdar_ctx *ctx, void *image, size_t bytes_per_rank, hipStream_t stream)
{
    IMPDAR_ARG_CHECK(ctx && ctx->comm, "communicator not initialised (impdar_comm_init)");
    char *base = reinterpret_cast<char *>(image);
    IMPDAR_NCCL_CHECK(ncclAllGather(base + (size_t)ctx->rank * bytes_per_rank, base, bytes_per_rank, ncclChar,
                                    reinterpret_cast<ncclComm_t>(ctx->comm), stream));
    return IMPDAR_OK;
}

// Grouped point-to-point exchange on one buffer: this rank sends bytes [soff[i], soff[i] + slen[i]) of `image` to
// rank speer[i] and receives bytes [roff[i], roff[i] + rlen[i]) from rank rpeer[i] (the halo exchange of the
// sharded Kirchhoff migration: every byte range is a run of whole image rows).  One ncclGroup, so the
// transfers of all peers proceed concurrently over their own xGMI links.
int impdar_exchange_ranges(impdar_ctx *ctx, void *image, int nsend, const int *speer, const size_t *soff,
                           const size_t *slen, int nrecv, const int *rpeer, const size_t *roff, const size_t *rlen,
                           hipStream_t stream)
{
    IMPDAR_ARG_CHECK(ctx && ctx->comm, "communicator not initialised (impdar_comm_init)");
    char *base = reinterpret_cast<char *>(image);
    ncclComm_t comm = reinterpret_cast<ncclComm_t>(ctx->comm);
    for (int i = 0; i < nsend; ++i)
        IMPDAR_ARG_CHECK(speer[i] >= 0 && speer[i] < ctx->nranks, "send peer %d outside the communicator", speer[i]);
    for (int i = 0; i < nrecv; ++i)
        IMPDAR_ARG_CHECK(rpeer[i] >= 0 && rpeer[i] < ctx->nranks, "receive peer %d outside the communicator", rpeer[i]);
    IMPDAR_NCCL_CHECK(ncclGroupStart());
    ncclResult_t bad = ncclSuccess;
    for (int i = 0; i < nsend && bad == ncclSuccess; ++i)
        if (slen[i]) bad = ncclSend(base + soff[i], slen[i], ncclChar, speer[i], comm, stream);
    for (int i = 0; i < nrecv && bad == ncclSuccess; ++i)
        if (rlen[i]) bad = ncclRecv(base + roff[i], rlen[i], ncclChar, rpeer[i], comm, stream);
    ncclResult_t end = ncclGroupEnd();
    if (bad != ncclSuccess || end != ncclSuccess) {
        impdar_set_error("grouped ncclSend/ncclRecv failed: %s", ncclGetErrorString(bad != ncclSuccess ? bad : end));
        return IMPDAR_ERR_COMM;
    }
    return IMPDAR_OK;
}

// Grouped point-to-point exchange between two buffers (the all-to-all of the kx-sharded phase shift): bytes
// [soff[i], soff[i] + slen[i]) of `sendbuf` go to rank peer[i], bytes [roff[i], roff[i] + rlen[i]) of `recvbuf` come
// from it; the block for this rank itself travels through RCCL as well (a device-local copy inside the group).
int impdar_exchange_buffers(impdar_ctx *ctx, const void *sendbuf, void *recvbuf, int npeer, const int *peer, const size_t *soff,
                            const size_t *slen, const size_t *roff, const size_t *rlen, hipStream_t stream)
{
    IMPDAR_ARG_CHECK(ctx && ctx->comm, "communicator not initialised (impdar_comm_init)");
    ncclComm_t comm = reinterpret_cast<ncclComm_t>(ctx->comm);
    for (int i = 0; i < npeer; ++i)
        IMPDAR_ARG_CHECK(peer[i] >= 0 && peer[i] < ctx->nranks, "peer %d outside the communicator", peer[i]);
    IMPDAR_NCCL_CHECK(ncclGroupStart());
    ncclResult_t bad = ncclSuccess;
    for (int i = 0; i < npeer && bad == ncclSuccess; ++i)
        if (slen[i]) bad = ncclSend(reinterpret_cast<const char *>(sendbuf) + soff[i], slen[i], ncclChar, peer[i], comm, stream);
    for (int i = 0; i < npeer && bad == ncclSuccess; ++i)
        if (rlen[i]) bad = ncclRecv(reinterpret_cast<char *>(recvbuf) + roff[i], rlen[i], ncclChar, peer[i], comm, stream);
    ncclResult_t end = ncclGroupEnd();
    if (bad != ncclSuccess || end != ncclSuccess) {
        impdar_set_error("grouped ncclSend/ncclRecv failed: %s", ncclGetErrorString(bad != ncclSuccess ? bad : end));
        return IMPDAR_ERR_COMM;
    }
    return IMPDAR_OK;
}

extern "C" int impdar_comm_barrier(impdar_ctx *ctx)
{
    IMPDAR_ARG_CHECK(ctx, "null context");
    IMPDAR_HIP_CHECK(hipSetDevice(ctx->device));
    if (ctx->comm && ctx->nranks > 1) {
        static thread_local DevBuf scratch;
        IMPDAR_HIP_CHECK(scratch.ensure(64));
        IMPDAR_NCCL_CHECK(ncclAllReduce(scratch.p, scratch.p, 1, ncclInt, ncclSum,
                                        reinterpret_cast<ncclComm_t>(ctx->comm), ctx->stream));
    }
    IMPDAR_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return IMPDAR_OK;
}
