// rccl_pair.hip -- include/rbp_rccl.h: the exchange hook of a sharded joint solve (rbp_session_shard_joint) as RCCL send / recv over xGMI.
// A library of its own (lib/librbp_rccl.so) so that the core library does not depend on RCCL.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstring>
#include <string>
#include <thread>

#include "rbp_rccl.h"

static_assert(sizeof(ncclUniqueId) == RBP_RCCL_ID_BYTES, "RBP_RCCL_ID_BYTES must be sizeof(ncclUniqueId)");

struct rbp_rccl_pair {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int device = 0, rank = 0, peer = 0;
    double timeout_s = 300.0;  // an exchange that does not complete within this many seconds aborts the communicator (<= 0: wait for ever)
    bool aborted = false;
};

namespace {
thread_local std::string g_err;
int fail(const std::string& what) {
    g_err = what;
    return 1;
}
// the pair is unusable after a failed exchange (the two ranks' send / recv sequences no longer match): abort the communicator so that
// neither this rank nor a peer blocked in the matching receive waits for ever
int abort_pair(rbp_rccl_pair* p, const std::string& what) {
    if (p && p->comm && !p->aborted) {
        (void)ncclCommAbort(p->comm);
        p->comm = nullptr, p->aborted = true;
    }
    return fail(what);
}
}  // namespace

extern "C" {

const char* rbp_rccl_last_error(void) { return g_err.c_str(); }

int rbp_rccl_unique_id(void* id_out) {
    if (!id_out) return fail("rbp_rccl_unique_id: null buffer");
    ncclUniqueId id;
    const ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return fail(std::string("ncclGetUniqueId: ") + ncclGetErrorString(r));
    std::memcpy(id_out, &id, sizeof(id));
    return 0;
}

int rbp_rccl_pair_create(rbp_rccl_pair** out, int device, int rank, int nranks, const void* id_bytes) {
    if (!out || !id_bytes) return fail("rbp_rccl_pair_create: null argument");
    if (!(nranks == 2 || nranks == 1) || rank < 0 || rank >= nranks) return fail("rbp_rccl_pair_create: a pair has two ranks (or one: the self-test), rank in range");
    if (hipSetDevice(device) != hipSuccess) return fail("rbp_rccl_pair_create: hipSetDevice failed");
    rbp_rccl_pair* p = new rbp_rccl_pair();
    p->device = device, p->rank = rank, p->peer = nranks == 2 ? 1 - rank : rank;
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof(id));
    const ncclResult_t r = ncclCommInitRank(&p->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        delete p;
        return fail(std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
    }
    if (hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) != hipSuccess) {
        ncclCommDestroy(p->comm);
        delete p;
        return fail("rbp_rccl_pair_create: hipStreamCreate failed");
    }
    *out = p;
    return 0;
}

int rbp_rccl_exchange(void* pair, void* send_dev, void* recv_dev, size_t bytes) {
    rbp_rccl_pair* p = static_cast<rbp_rccl_pair*>(pair);
    if (!p || !send_dev || !recv_dev) return fail("rbp_rccl_exchange: null argument");
    if (p->aborted || !p->comm) return fail("rbp_rccl_exchange: the pair's communicator was aborted by an earlier failure");
    if (bytes == 0) return 0;
    ncclResult_t r = ncclGroupStart();
    if (r == ncclSuccess) r = ncclSend(send_dev, bytes, ncclChar, p->peer, p->comm, p->stream);
    if (r == ncclSuccess) r = ncclRecv(recv_dev, bytes, ncclChar, p->peer, p->comm, p->stream);
    const ncclResult_t e = ncclGroupEnd();
    if (r == ncclSuccess) r = e;
    if (r != ncclSuccess) return abort_pair(p, std::string("rbp_rccl_exchange: ") + ncclGetErrorString(r));
    // wait for the transfer WITHOUT blocking for ever on a peer that is gone: poll the stream, the communicator's asynchronous error and a
    // clock; on either, ncclCommAbort -- which also releases a peer blocked in its own receive -- and report failure (rbp_session_run then
    // returns RBP_ERR_EXCHANGE on this rank; the peer's hook fails the same way: both ranks abort together)
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; ++spin) {
        const hipError_t q = hipStreamQuery(p->stream);
        if (q == hipSuccess) return 0;
        if (q != hipErrorNotReady) return abort_pair(p, std::string("rbp_rccl_exchange: stream error: ") + hipGetErrorString(q));
        ncclResult_t ae = ncclSuccess;
        if (ncclCommGetAsyncError(p->comm, &ae) != ncclSuccess || (ae != ncclSuccess && ae != ncclInProgress))
            return abort_pair(p, std::string("rbp_rccl_exchange: asynchronous communicator error: ") + ncclGetErrorString(ae));
        if (p->timeout_s > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > p->timeout_s)
            return abort_pair(p, "rbp_rccl_exchange: timed out waiting for the peer rank (rbp_rccl_pair_set_timeout)");
        if (spin > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));  // (the first ~ms spins: an exchange is 20 us .. 1 ms)
    }
}

// the stream-ordered form (rbp_exchange_stream_fn): the grouped send / receive is enqueued on the CALLER's stream and the call returns
int rbp_rccl_exchange_stream(void* pair, void* send_dev, void* recv_dev, size_t bytes, void* stream) {
    rbp_rccl_pair* p = static_cast<rbp_rccl_pair*>(pair);
    if (!p || !send_dev || !recv_dev) return fail("rbp_rccl_exchange_stream: null argument");
    if (p->aborted || !p->comm) return fail("rbp_rccl_exchange_stream: the pair's communicator was aborted by an earlier failure");
    if (bytes == 0) return 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    ncclResult_t r = ncclGroupStart();
    if (r == ncclSuccess) r = ncclSend(send_dev, bytes, ncclChar, p->peer, p->comm, st);
    if (r == ncclSuccess) r = ncclRecv(recv_dev, bytes, ncclChar, p->peer, p->comm, st);
    const ncclResult_t e = ncclGroupEnd();
    if (r == ncclSuccess) r = e;
    if (r != ncclSuccess) return abort_pair(p, std::string("rbp_rccl_exchange_stream: ") + ncclGetErrorString(r));
    return 0;
}
// rbp_exchange_abort_fn: what rbp_session_shard_joint_stream calls when a round of the exchange has timed out
int rbp_rccl_abort(void* pair) {
    rbp_rccl_pair* p = static_cast<rbp_rccl_pair*>(pair);
    if (!p) return fail("rbp_rccl_abort: null pair");
    (void)abort_pair(p, "rbp_rccl_abort: aborted on request");
    return 0;
}

int rbp_rccl_pair_set_timeout(rbp_rccl_pair* p, double seconds) {
    if (!p) return fail("rbp_rccl_pair_set_timeout: null pair");
    p->timeout_s = seconds;
    return 0;
}

void rbp_rccl_pair_destroy(rbp_rccl_pair* p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    if (p->comm) (void)ncclCommDestroy(p->comm);
    delete p;
}

}  // extern "C"
