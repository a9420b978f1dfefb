// Peer-to-peer all-reduce of the matcher's two tiny records (SURVEY.md §5 "Distributed communication backend", §8e): the
// correspondence counts (2 x int32) and the Gram record (72 x f64) of the sharded scan-to-map iteration
// (lili_s2m_iterate_sharded; the loop it shards is L/src/BackendFusion.cpp:1536,1606).  A ring or tree collective is the wrong
// tool for < 1 KB: the exchange is pure latency.  Here every rank owns a MAILBOX in its own HBM (fine-grained, mapped into every
// peer through hipIpc), with one slot per source rank; ONE small kernel per all-reduce
//   1. stores this rank's record into slot[rank] of every peer's mailbox (xGMI stores, one hop, all peers at once),
//   2. drains its store counter, then writes the slot's sequence flag,
//   3. waits until all `world` flags of its OWN mailbox carry the sequence number,
//   4. adds the `world` records IN RANK ORDER into the caller's buffer
// so every rank computes the identical sum bit for bit (a ring all-reduce gives each rank a different association order; here
// no pose broadcast is needed afterwards and the ranks cannot drift apart).  lili_p2p_allreduce has the signature of
// ncclAllReduce and is handed to lili_s2m_iterate_sharded in its place.
//
// Slots are double-buffered by the parity of the sequence number: a rank can run at most one all-reduce ahead of the slowest
// rank (call k+1 needs every rank's contribution to call k+1, which a rank sends only after its own call k has completed), so
// the slot written for call k+2 (parity of k) is never still being read.  A wait that does not complete within the communicator's
// timeout (lili_p2p_set_timeout, default 10 s of device time) gives up instead of hanging the GPU: the status word in pinned host memory
// is raised (lili_p2p_status), the failure word of this rank's AND of every peer's mailbox is set, the rank publishes nothing any more,
// and every later exchange on any rank fails at its first look — lili_s2m_iterate_sharded then reports LILI_E_STATE.
#include "lili_ctx.h"
#include "lili_p2p_dev.h"

#include <new>

namespace {

// generic form: sendbuff / recvbuff hold `count` elements of 4 (int32) or 8 (f64) bytes; one wave, W words per lane
template <int W>
__global__ __launch_bounds__(64) void k_p2p_allreduce(const void* __restrict__ send, void* __restrict__ recv, int count, int elem8, P2PView v) {
    const int t = threadIdx.x;
    const unsigned long long was_dead = p2p_dead_word(v);
    unsigned long long w[W], s[W];
#pragma unroll
    for (int i = 0; i < W; i++) { w[i] = 0ull; s[i] = 0ull; }
    if (elem8) {
#pragma unroll
        for (int i = 0; i < W; i++) if (t + 64 * i < count) w[i] = reinterpret_cast<const unsigned long long*>(send)[t + 64 * i];
        if (!p2p_exchange_words<true, W>(v, count, w, s, was_dead)) return;
#pragma unroll
        for (int i = 0; i < W; i++) if (t + 64 * i < count) reinterpret_cast<unsigned long long*>(recv)[t + 64 * i] = s[i];
    } else {
#pragma unroll
        for (int i = 0; i < W; i++) if (t + 64 * i < count) w[i] = (unsigned long long)(unsigned)reinterpret_cast<const int*>(send)[t + 64 * i];
        if (!p2p_exchange_words<false, W>(v, count, w, s, was_dead)) return;
#pragma unroll
        for (int i = 0; i < W; i++) if (t + 64 * i < count) reinterpret_cast<int*>(recv)[t + 64 * i] = (int)(unsigned)s[i];
    }
}

}  // namespace

struct lili_p2p {
    lili_ctx* ctx = nullptr;
    int rank = 0, world = 1;
    unsigned long long* box = nullptr;        // my mailbox (device, fine-grained)
    void* peer_map[kP2PMaxWorld] = {};        // hipIpcOpenMemHandle results (nullptr for myself)
    P2PView view{};
    bool connected = false;
    unsigned long long seq = 0;
    int* status = nullptr;                    // pinned host word: 0 ok, 1 = a wait timed out
    double timeout_s = 10.0;                  // lili_p2p_set_timeout
};

extern "C" {

int lili_p2p_create(lili_ctx* ctx, int rank, int world, lili_p2p** out) {
    if (!ctx || !out) return LILI_E_ARG;
    *out = nullptr;
    ARGCHK(world >= 1 && world <= kP2PMaxWorld && rank >= 0 && rank < world, "p2p_create: bad rank / world (at most 16 ranks)");
    HIPCHK(hipSetDevice(ctx->device));
    lili_p2p* c = new (std::nothrow) lili_p2p();
    if (!c) return LILI_E_NOMEM;
    c->ctx = ctx; c->rank = rank; c->world = world;
    const size_t bytes = (size_t)2 * kP2PMaxWorld * kP2PSlotBytes + 128;      // slots + the sticky failure word (kP2PDeadWord)
    void* p = nullptr;
    // uncached / fine-grained: stores of a peer GPU must become visible to a kernel that is already running
    hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained); }
    if (e != hipSuccess) { delete c; return ctx->fail(LILI_E_HIP, std::string("p2p_create: fine-grained allocation failed: ") + hipGetErrorString(e)); }
    c->box = reinterpret_cast<unsigned long long*>(p);
    if (hipMemset(p, 0, bytes) != hipSuccess || hipHostMalloc(reinterpret_cast<void**>(&c->status), sizeof(int), hipHostMallocDefault) != hipSuccess) {
        (void)hipFree(p); delete c;
        return ctx->fail(LILI_E_HIP, "p2p_create: allocation failed");
    }
    *c->status = 0;
    c->view.box[rank] = c->box; c->view.rank = rank; c->view.world = world; c->view.status = c->status;
    c->view.timeout_ticks = (long long)(c->timeout_s * 1e8);
    if (world == 1) c->connected = true;
    *out = c;
    return LILI_OK;
}

int lili_p2p_handle(lili_p2p* c, void* handle /*LILI_P2P_HANDLE_BYTES*/) {
    if (!c || !handle) return LILI_E_ARG;
    lili_ctx* ctx = c->ctx;
    static_assert(sizeof(hipIpcMemHandle_t) <= LILI_P2P_HANDLE_BYTES, "handle size");
    hipIpcMemHandle_t h;
    HIPCHK(hipSetDevice(ctx->device));
    HIPCHK(hipIpcGetMemHandle(&h, c->box));
    std::memset(handle, 0, LILI_P2P_HANDLE_BYTES);
    std::memcpy(handle, &h, sizeof(h));
    return LILI_OK;
}

int lili_p2p_connect(lili_p2p* c, const void* all_handles /*world x LILI_P2P_HANDLE_BYTES, rank order*/) {
    if (!c || !all_handles) return LILI_E_ARG;
    lili_ctx* ctx = c->ctx;
    HIPCHK(hipSetDevice(ctx->device));
    for (int r = 0; r < c->world; r++) {
        if (r == c->rank) continue;
        hipIpcMemHandle_t h;
        std::memcpy(&h, reinterpret_cast<const char*>(all_handles) + (size_t)r * LILI_P2P_HANDLE_BYTES, sizeof(h));
        void* p = nullptr;
        HIPCHK(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        c->peer_map[r] = p;
        c->view.box[r] = reinterpret_cast<unsigned long long*>(p);
    }
    c->connected = true;
    return LILI_OK;
}

// lili_allreduce_fn: in place or out of place, datatype 2 (int32) / 8 (f64), op 0 (sum); comm = lili_p2p*
int lili_p2p_allreduce(const void* sendbuff, void* recvbuff, size_t count, int datatype, int op, void* comm, void* stream) {
    lili_p2p* c = reinterpret_cast<lili_p2p*>(comm);
    if (!c || !c->connected || !sendbuff || !recvbuff || op != 0 || (datatype != 2 && datatype != 8) || count == 0 || count > (size_t)kP2PMaxCount) return 1;
    if (hipSetDevice(c->ctx->device) != hipSuccess) return 1;
    P2PView v = c->view;
    v.seq = ++c->seq;
    if (count <= 128) hipLaunchKernelGGL(k_p2p_allreduce<2>, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), sendbuff, recvbuff, (int)count, datatype == 8 ? 1 : 0, v);
    else hipLaunchKernelGGL(k_p2p_allreduce<kP2PMaxPerLane>, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), sendbuff, recvbuff, (int)count, datatype == 8 ? 1 : 0, v);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int lili_p2p_status(lili_p2p* c) { return c && c->status ? *c->status : -1; }

// How long an exchange waits for a peer's record before it gives up (device time; default 10 s).  The first exchange of a communicator
// has to absorb whatever skew the ranks start with — lazy code-object loads, data loading, a host stall: keep this generous or put a
// barrier of the control plane in front of the first lili_s2m_iterate_sharded.  Takes effect with the next exchange.
int lili_p2p_set_timeout(lili_p2p* c, double seconds) {
    if (!c || !(seconds > 0.0) || !(seconds <= 3600.0)) return LILI_E_ARG;
    c->timeout_s = seconds;
    c->view.timeout_ticks = (long long)(seconds * 1e8);
    return LILI_OK;
}

void lili_p2p_destroy(lili_p2p* c) {
    if (!c) return;
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->ctx->stream);
    for (int r = 0; r < c->world; r++) if (c->peer_map[r]) (void)hipIpcCloseMemHandle(c->peer_map[r]);
    if (c->box) (void)hipFree(c->box);
    if (c->status) (void)hipHostFree(c->status);
    delete c;
}

}  // extern "C"

// internal (lili_match.hip): the view of the NEXT exchange of this communicator (advances its sequence number)
lili::P2PView lili_p2p_next_view(lili_p2p* c) {
    P2PView v = c->view;
    v.seq = ++c->seq;
    return v;
}
bool lili_p2p_usable(const lili_p2p* c, const lili_ctx* ctx) { return c && c->connected && c->ctx == ctx; }

