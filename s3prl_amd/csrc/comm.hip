// comm.hip — the data-parallel exchange of the path behind the C ABI (include/s3enc.h "multi-GPU"):
// one process (or thread) per GPU, utterances sharded in contiguous blocks, every rank's (states, shard, T, D) slab
// re-assembled into (states, world * shard, T, D) with ONE RCCL all-gather per state, so that hidden_states[l] comes out as a
// contiguous (B, T, D) block in rank order (a single flat gather would be rank-major across states).  The gather of state l
// is ordered after the event the encoder records when that state is final (s3enc_set_layer_events) and runs on the
// communicator's own stream: all but the last gather overlap the remaining layers' compute.
//
// RCCL is reached through dlopen("librccl.so.1"): a process that already loaded a copy (PyTorch ships one) keeps using that
// one, a plain C / Go / Rust binder gets /opt/rocm/lib's, and libs3enc.so has no link-time dependency on it.  xGMI is
// point-to-point (7 links per GPU): RCCL picks the ring / direct algorithm; nothing here assumes a switch.
#include <dlfcn.h>

#include "engine_internal.h"

namespace {

typedef int ncclResult_t;
typedef struct ncclComm* ncclComm_t;
struct ncclUniqueId {
    char internal[128];
};
enum { ncclInt8 = 0 };  // an all-gather only moves bytes

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

Rccl load_rccl() {
    Rccl r;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (r.lib) break;
    }
    if (!r.lib) {
        const char* e = dlerror();  // ONE call: dlerror() clears the error it returns
        r.why = std::string("librccl.so.1 not found: ") + (e ? e : "");
        return r;
    }
    auto sym = [&](const char* s) {
        void* p = dlsym(r.lib, s);
        if (!p && r.why.empty()) r.why = std::string("librccl has no symbol ") + s;
        return p;
    };
    r.GetVersion = (decltype(r.GetVersion))sym("ncclGetVersion");
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    return r;
}

// one rank per GPU may be a THREAD: the first calls of two ranks can race, so the table is a C++11 magic static
Rccl& rccl() {
    static Rccl r = load_rccl();
    return r;
}

}  // namespace

struct s3enc_comm_s {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0, device = 0;
    hipStream_t stream = nullptr;  // the communicator's own stream: gathers overlap the encoder's stream
    hipEvent_t done = nullptr;     // recorded after the last gather of a call; the caller's stream waits for it
};

#define RCCL_TRY(expr)                                                                                       \
    do {                                                                                                     \
        ncclResult_t _r = (expr);                                                                            \
        if (_r != 0) {                                                                                       \
            char _b[512];                                                                                    \
            snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr, R.GetErrorString ? R.GetErrorString(_r) : "?", \
                     __FILE__, __LINE__);                                                                    \
            return fail(_b);                                                                                 \
        }                                                                                                    \
    } while (0)

extern "C" {

int s3enc_comm_version(int32_t* version) {
    Rccl& R = rccl();
    if (!R.lib || !R.why.empty()) return fail("s3enc_comm: " + R.why);
    if (!version) return fail("s3enc_comm_version: null argument");
    int v = 0;
    RCCL_TRY(R.GetVersion(&v));
    *version = v;
    return 0;
}

int s3enc_comm_unique_id(void* id128) {
    Rccl& R = rccl();
    if (!R.lib || !R.why.empty()) return fail("s3enc_comm: " + R.why);
    if (!id128) return fail("s3enc_comm_unique_id: null argument");
    ncclUniqueId id;
    RCCL_TRY(R.GetUniqueId(&id));
    memcpy(id128, id.internal, sizeof(id.internal));
    return 0;
}

int s3enc_comm_init_rank(const void* id128, int32_t world, int32_t rank, int32_t device, s3enc_comm* out) {
    Rccl& R = rccl();
    if (!R.lib || !R.why.empty()) return fail("s3enc_comm: " + R.why);
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return fail("s3enc_comm_init_rank: bad arguments");
    DeviceGuard dg(device);
    if (!dg.ok) return fail("s3enc_comm_init_rank: hipSetDevice failed");
    s3enc_comm_s* c = new s3enc_comm_s();
    c->world = world;
    c->rank = rank;
    c->device = device;
    ncclUniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    ncclResult_t r = R.CommInitRank(&c->comm, world, id, rank);
    if (r != 0) {
        std::string msg = std::string("ncclCommInitRank failed: ") + (R.GetErrorString ? R.GetErrorString(r) : "?");
        delete c;
        return fail(msg);
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) {
        (void)R.CommDestroy(c->comm);
        delete c;
        return fail("s3enc_comm_init_rank: stream / event creation failed");
    }
    *out = c;
    return 0;
}

int s3enc_comm_destroy(s3enc_comm c) {
    if (!c) return 0;
    Rccl& R = rccl();
    DeviceGuard dg(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->comm && R.CommDestroy) (void)R.CommDestroy(c->comm);
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

int s3enc_comm_info(s3enc_comm c, int32_t* world, int32_t* rank) {
    if (!c || !world || !rank) return fail("s3enc_comm_info: null argument");
    *world = c->world;
    *rank = c->rank;
    return 0;
}

// algo S3ENC_EXCHANGE_COLLECTIVE: one ncclAllGather per state (RCCL chooses ring / tree / direct by its own tuning);
// algo S3ENC_EXCHANGE_DIRECT: per state ONE group of world-1 ncclSend + world-1 ncclRecv, peer (rank +- p) mod world at step p —
// every pair of GPUs owns its xGMI link (7 links x ~153 GB/s per GPU), so the all-pairs form drives all seven links at
// once where a ring collective is bound by one (cfg4 bf16 at N = 8: 5.72 GB inbound per GPU = >= 5.3 ms direct against ~37 ms
// through a single-link ring).  Both forms produce the same bytes in the same places.
int s3enc_comm_exchange_states(s3enc_comm c, int32_t algo, const void* send, int64_t send_state_stride, void* recv,
                               int64_t recv_state_stride, int32_t n_states, int64_t bytes_per_state, void* const* ready_events,
                               void* stream) {
    Rccl& R = rccl();
    if (!R.lib || !R.why.empty()) return fail("s3enc_comm: " + R.why);
    if (!c || !send || !recv || n_states < 1 || bytes_per_state < 1) return fail("s3enc_comm_exchange_states: bad arguments");
    if (algo != S3ENC_EXCHANGE_COLLECTIVE && algo != S3ENC_EXCHANGE_DIRECT) return fail("s3enc_comm_exchange_states: unknown algo");
    if (recv_state_stride < bytes_per_state * c->world || send_state_stride < bytes_per_state)
        return fail("s3enc_comm_exchange_states: a state stride is smaller than the block it holds");
    DeviceGuard dg(c->device);
    hipStream_t caller = (hipStream_t)stream;
    if (!ready_events) {  // no per-state events: the gathers simply follow everything enqueued on the caller's stream so far
        HIP_TRY(hipEventRecord(c->done, caller));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->done, 0));
    } else {
        for (int l = 0; l < n_states; ++l)
            if (!ready_events[l]) return fail("s3enc_comm_exchange_states: ready_events holds fewer than n_states events");
    }
    for (int l = 0; l < n_states; ++l) {
        if (ready_events) HIP_TRY(hipStreamWaitEvent(c->stream, (hipEvent_t)ready_events[l], 0));
        const char* src = (const char*)send + (size_t)l * send_state_stride;
        char* dst = (char*)recv + (size_t)l * recv_state_stride;
        if (algo == S3ENC_EXCHANGE_COLLECTIVE) {
            RCCL_TRY(R.AllGather(src, dst, (size_t)bytes_per_state, ncclInt8, c->comm, c->stream));
            continue;
        }
        char* own = dst + (size_t)c->rank * bytes_per_state;
        // the rank's own block: a device copy — or, under the tuning key comm_self_p2p, one more send / receive pair of the
        // state's group with peer = rank (pstep 0), which is how the all-pairs code below runs on a one-GPU box
        const bool self_p2p = s3::tuning().comm_self_p2p != 0 && own != src;
        if (own != src && !self_p2p) HIP_TRY(hipMemcpyAsync(own, src, (size_t)bytes_per_state, hipMemcpyDeviceToDevice, c->stream));
        if (c->world == 1 && !self_p2p) continue;
        RCCL_TRY(R.GroupStart());
        for (int pstep = self_p2p ? 0 : 1; pstep < c->world; ++pstep) {
            const int to = (c->rank + pstep) % c->world, from = (c->rank - pstep + c->world) % c->world;
            ncclResult_t rs = R.Send(src, (size_t)bytes_per_state, ncclInt8, to, c->comm, c->stream);
            ncclResult_t rr = rs ? rs : R.Recv(dst + (size_t)from * bytes_per_state, (size_t)bytes_per_state, ncclInt8, from, c->comm, c->stream);
            if (rr != 0) {
                (void)R.GroupEnd();
                return fail(std::string("ncclSend / ncclRecv failed: ") + (R.GetErrorString ? R.GetErrorString(rr) : "?"));
            }
        }
        RCCL_TRY(R.GroupEnd());
    }
    HIP_TRY(hipEventRecord(c->done, c->stream));
    HIP_TRY(hipStreamWaitEvent(caller, c->done, 0));  // the caller's later work sees the gathered states
    return 0;
}

int s3enc_comm_allgather_states(s3enc_comm c, const void* send, int64_t send_state_stride, void* recv, int64_t recv_state_stride,
                                int32_t n_states, int64_t bytes_per_state, void* const* ready_events, void* stream) {
    return s3enc_comm_exchange_states(c, S3ENC_EXCHANGE_COLLECTIVE, send, send_state_stride, recv, recv_state_stride, n_states,
                                      bytes_per_state, ready_events, stream);
}

}  // extern "C"
