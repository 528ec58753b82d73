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
#include <unistd.h>

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

// ---- S3ENC_EXCHANGE_COPY (round 6): the exchange on the copy engines ------------------------------------------------------------
// Every rank owns ONE receive slab (registered once) and a mailbox of 2 x world 64-bit sequence numbers; both are exported by
// hipIpcGetMemHandle and mapped by every peer.  Exchange number q (1, 2, ...; every rank issues the same sequence of calls):
//   ack   rank r writes q into peer p's mailbox.acks[r]: "the consumers of my slab's previous contents are behind me on my stream,
//         you may write exchange q into it";
//   push  per state, on the stream rank r keeps for peer p and behind the encoder's "state l final" event: ONE hipMemcpyAsync of
//         rank r's block into peer p's slab at [l][r] — peer-to-peer copies run on the SDMA engines, one stream per peer so that all
//         world - 1 xGMI links are driven at once, and no compute unit multiplies anything for it;
//   flag  behind rank r's last push to p: q into p's mailbox.flags[r];
//   wait  rank r's own stream waits until mailbox.flags[p] >= q for every peer p, then the caller's stream is ordered behind it.
// The two waits (for an ack before the first push, for the flags at the end) are one-wave kernels that poll with a deadline; a
// deadline that passes sets the communicator's error word (s3enc_comm_copy_status) instead of hanging the GPU.  Sequence numbers only
// grow, so nothing is ever reset and a slow rank cannot be overtaken by more than the one exchange the ack protocol admits.
struct CopyPeer {
    char* slab = nullptr;        // peer's receive slab, mapped here
    uint64_t* mbox = nullptr;    // peer's mailbox, mapped here: flags[world] | acks[world]
    void* slab_map = nullptr;    // what hipIpcOpenMemHandle returned (the allocation's base)
    void* mbox_map = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t sent = nullptr;
};
struct CopyHandle {  // what travels between the ranks (S3ENC_COPY_HANDLE_BYTES)
    hipIpcMemHandle_t slab, mbox;
    uint64_t slab_offset, slab_bytes, mbox_offset;
    int32_t rank, pid;
    char pad[S3ENC_COPY_HANDLE_BYTES - 2 * sizeof(hipIpcMemHandle_t) - 3 * sizeof(uint64_t) - 2 * sizeof(int32_t)];
};
static_assert(sizeof(CopyHandle) == S3ENC_COPY_HANDLE_BYTES, "CopyHandle layout");

struct s3enc_comm_s {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0, device = 0;
    hipStream_t stream = nullptr;  // the communicator's own stream: gathers overlap the encoder's stream
    hipEvent_t done = nullptr;     // recorded after the last gather of a call; the caller's stream waits for it
    // S3ENC_EXCHANGE_COPY
    uint64_t* mbox = nullptr;      // mine: flags[world] | acks[world] | error word
    char* slab = nullptr;          // my registered receive slab
    int64_t slab_bytes = 0;
    uint64_t seq = 0;
    bool attached = false;
    hipEvent_t free_ev = nullptr;  // s3enc_comm_copy_release: "the slab's last readers are in front of this point of the caller's stream"
    bool have_free = false;
    std::vector<CopyPeer> peers;
};

namespace {
__global__ void copy_put_kernel(uint64_t* dst, uint64_t v) {
    __threadfence_system();
    __hip_atomic_store(dst, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// lane `only` — or, with only < 0, each of the lanes 0 .. n - 1 but `skip` — waits for flags[lane] >= v; `ticks` reference-clock
// ticks at most
__global__ void copy_wait_kernel(const uint64_t* flags, int n, int skip, int only, uint64_t v, uint64_t* err, long long ticks) {
    const int i = threadIdx.x;
    if (only >= 0 ? i != only : (i >= n || i == skip)) return;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(&flags[i], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < v) {
        __builtin_amdgcn_s_sleep(16);
        if (wall_clock64() - t0 > ticks) {
            atomicOr((unsigned long long*)err, 1ull << (i & 31));
            return;
        }
    }
}
long long copy_deadline_ticks(int dev) {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;
    const char* e = getenv("S3ENC_COPY_DEADLINE_MS");
    const double ms = e ? atof(e) : 5000.0;
    return (long long)((ms < 1.0 ? 1.0 : (ms > 60000.0 ? 60000.0 : ms)) * khz);
}
}  // namespace

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

static int copy_exchange(s3enc_comm c, const void* send, int64_t send_state_stride, void* recv, int64_t recv_state_stride, int32_t n_states,
                         int64_t bytes_per_state, void* const* ready_events, void* stream) {
    if (recv_state_stride < bytes_per_state * c->world || send_state_stride < bytes_per_state)
        return fail("s3enc_comm_exchange_states: a state stride is smaller than the block it holds");
    if (c->world > 1) {
        if (!c->attached) return fail("S3ENC_EXCHANGE_COPY: s3enc_comm_copy_export / s3enc_comm_copy_attach first");
        if ((char*)recv != c->slab) return fail("S3ENC_EXCHANGE_COPY: recv is not the registered receive slab (the peers write through their mapping of it)");
        if ((int64_t)(n_states - 1) * recv_state_stride + (int64_t)c->world * bytes_per_state > c->slab_bytes)
            return fail("S3ENC_EXCHANGE_COPY: the exchange does not fit the registered receive slab");
    }
    if (ready_events)
        for (int l = 0; l < n_states; ++l)
            if (!ready_events[l]) return fail("s3enc_comm_exchange_states: ready_events holds fewer than n_states events");
    DeviceGuard dg(c->device);
    hipStream_t caller = (hipStream_t)stream;
    const uint64_t q = ++c->seq;
    const long long ticks = copy_deadline_ticks(c->device);
    // Two gates.  `slab_free`: the last readers of the slab's previous contents — in front of the acks and of my own block's copy.
    // With s3enc_comm_copy_release the caller marked that point BEFORE it enqueued this step's forward, so the pushes of state l can
    // start behind "state l final" while the later layers still compute; without it the point is "now" (the whole forward the caller
    // enqueued before this call is in front of the exchange: correct, nothing overlaps).  `here`: everything enqueued so far — what the
    // pushes wait for when there are no per-state events.
    HIP_TRY(hipEventRecord(c->done, caller));
    hipEvent_t here = c->done, slab_free = c->have_free ? c->free_ev : c->done;
    c->have_free = false;
    HIP_TRY(hipStreamWaitEvent(c->stream, slab_free, 0));
    if (!ready_events) HIP_TRY(hipStreamWaitEvent(c->stream, here, 0));
    uint64_t* err = c->mbox ? c->mbox + 2 * c->world : nullptr;
    for (int pi = 0; pi < c->world; ++pi) {
        if (pi == c->rank) continue;
        CopyPeer& pr = c->peers[pi];
        HIP_TRY(hipStreamWaitEvent(pr.stream, slab_free, 0));
        if (!ready_events) HIP_TRY(hipStreamWaitEvent(pr.stream, here, 0));
        hipLaunchKernelGGL(copy_put_kernel, dim3(1), dim3(1), 0, pr.stream, pr.mbox + c->world + c->rank, q);  // ack: p may write exchange q here
        // ... and p must have said the same to me before my first block goes into ITS slab
        hipLaunchKernelGGL(copy_wait_kernel, dim3(1), dim3(64), 0, pr.stream, (const uint64_t*)(c->mbox + c->world), c->world, -1, pi, q, err, ticks);
    }
    for (int l = 0; l < n_states; ++l) {
        const char* src = (const char*)send + (size_t)l * send_state_stride;
        char* own = (char*)recv + (size_t)l * recv_state_stride + (size_t)c->rank * bytes_per_state;
        if (ready_events) HIP_TRY(hipStreamWaitEvent(c->stream, (hipEvent_t)ready_events[l], 0));
        if (own != src) HIP_TRY(hipMemcpyAsync(own, src, (size_t)bytes_per_state, hipMemcpyDeviceToDevice, c->stream));
        for (int pi = 0; pi < c->world; ++pi) {
            if (pi == c->rank) continue;
            CopyPeer& pr = c->peers[pi];
            if (ready_events) HIP_TRY(hipStreamWaitEvent(pr.stream, (hipEvent_t)ready_events[l], 0));
            HIP_TRY(hipMemcpyAsync(pr.slab + (size_t)l * recv_state_stride + (size_t)c->rank * bytes_per_state, src, (size_t)bytes_per_state,
                                   hipMemcpyDeviceToDevice, pr.stream));
        }
    }
    for (int pi = 0; pi < c->world; ++pi) {
        if (pi == c->rank) continue;
        CopyPeer& pr = c->peers[pi];
        hipLaunchKernelGGL(copy_put_kernel, dim3(1), dim3(1), 0, pr.stream, pr.mbox + c->rank, q);  // flag: my blocks of exchange q have landed
        HIP_TRY(hipEventRecord(pr.sent, pr.stream));
        HIP_TRY(hipStreamWaitEvent(c->stream, pr.sent, 0));  // `send` is free again once every push has read it
    }
    if (c->world > 1)
        hipLaunchKernelGGL(copy_wait_kernel, dim3(1), dim3(64), 0, c->stream, (const uint64_t*)c->mbox, c->world, c->rank, -1, q, err, ticks);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->done, c->stream));
    HIP_TRY(hipStreamWaitEvent(caller, c->done, 0));  // the caller's later work sees the gathered states
    return 0;
}

extern "C" {

// A communicator without an RCCL side: S3ENC_EXCHANGE_COPY only (the ranks meet through the IPC handles they exchange themselves).
int s3enc_comm_init_local(int32_t world, int32_t rank, int32_t device, s3enc_comm* out) {
    if (!out || world < 1 || world > 64 || rank < 0 || rank >= world) return fail("s3enc_comm_init_local: bad arguments (1 <= world <= 64)");
    DeviceGuard dg(device);
    if (!dg.ok) return fail("s3enc_comm_init_local: hipSetDevice failed");
    s3enc_comm_s* c = new s3enc_comm_s();
    c->world = world;
    c->rank = rank;
    c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) {
        delete c;
        return fail("s3enc_comm_init_local: stream / event creation failed");
    }
    *out = c;
    return 0;
}

// Register `recv_slab` (this rank's receive buffer of every later S3ENC_EXCHANGE_COPY exchange: (states, world * shard, T, D), device
// memory that stays allocated while the communicator lives) and write the S3ENC_COPY_HANDLE_BYTES the other ranks need to map it.
int s3enc_comm_copy_export(s3enc_comm c, void* recv_slab, int64_t bytes, void* handle_out) {
    if (!c || !recv_slab || bytes < 1 || !handle_out) return fail("s3enc_comm_copy_export: bad arguments");
    if (c->world > 64) return fail("s3enc_comm_copy_export: world <= 64");
    DeviceGuard dg(c->device);
    if (!c->mbox) {
        // the mailbox is polled by this GPU while its peers write it: uncached device memory (what RCCL uses for its flags), else
        // fine-grained, else plain
        const size_t mb = (size_t)(2 * c->world + 1) * sizeof(uint64_t);
        void* m = nullptr;
        if (hipExtMallocWithFlags(&m, mb, hipDeviceMallocUncached) != hipSuccess) {
            (void)hipGetLastError();
            if (hipExtMallocWithFlags(&m, mb, hipDeviceMallocFinegrained) != hipSuccess) {
                (void)hipGetLastError();
                HIP_TRY(hipMalloc(&m, mb));
            }
        }
        HIP_TRY(hipMemset(m, 0, mb));
        c->mbox = (uint64_t*)m;
    }
    CopyHandle h;
    memset(&h, 0, sizeof(h));
    void* base = nullptr;
    size_t span = 0;
    HIP_TRY(hipMemGetAddressRange((hipDeviceptr_t*)&base, &span, (hipDeviceptr_t)recv_slab));  // a framework's allocator hands out interior pointers
    if ((char*)recv_slab + bytes > (char*)base + span) return fail("s3enc_comm_copy_export: the slab runs past its allocation");
    HIP_TRY(hipIpcGetMemHandle(&h.slab, base));
    h.slab_offset = (uint64_t)((char*)recv_slab - (char*)base);
    h.slab_bytes = (uint64_t)bytes;
    void* mbase = nullptr;
    HIP_TRY(hipMemGetAddressRange((hipDeviceptr_t*)&mbase, &span, (hipDeviceptr_t)c->mbox));
    HIP_TRY(hipIpcGetMemHandle(&h.mbox, mbase));
    h.mbox_offset = (uint64_t)((char*)c->mbox - (char*)mbase);
    h.rank = c->rank;
    h.pid = (int32_t)getpid();
    c->slab = (char*)recv_slab;
    c->slab_bytes = bytes;
    memcpy(handle_out, &h, sizeof(h));
    return 0;
}

// `handles`: world x S3ENC_COPY_HANDLE_BYTES, entry r = what rank r's s3enc_comm_copy_export wrote (gathered by any side channel).
int s3enc_comm_copy_attach(s3enc_comm c, const void* handles) {
    if (!c || !handles) return fail("s3enc_comm_copy_attach: bad arguments");
    if (!c->slab || !c->mbox) return fail("s3enc_comm_copy_attach: s3enc_comm_copy_export first");
    if (c->attached) return fail("s3enc_comm_copy_attach: already attached");
    DeviceGuard dg(c->device);
    c->peers.assign(c->world, CopyPeer());
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank) continue;
        CopyHandle h;
        memcpy(&h, (const char*)handles + (size_t)r * S3ENC_COPY_HANDLE_BYTES, sizeof(h));
        if (h.rank != r) return fail("s3enc_comm_copy_attach: handle " + std::to_string(r) + " was exported by rank " + std::to_string(h.rank));
        if (h.pid == (int32_t)getpid()) return fail("s3enc_comm_copy_attach: ranks of one process cannot map each other through IPC handles (one process per GPU)");
        if ((int64_t)h.slab_bytes != c->slab_bytes) return fail("s3enc_comm_copy_attach: the ranks registered slabs of different sizes");
        CopyPeer& pr = c->peers[r];
        HIP_TRY(hipIpcOpenMemHandle(&pr.slab_map, h.slab, hipIpcMemLazyEnablePeerAccess));
        HIP_TRY(hipIpcOpenMemHandle(&pr.mbox_map, h.mbox, hipIpcMemLazyEnablePeerAccess));
        pr.slab = (char*)pr.slab_map + h.slab_offset;
        pr.mbox = (uint64_t*)((char*)pr.mbox_map + h.mbox_offset);
        HIP_TRY(hipStreamCreateWithFlags(&pr.stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&pr.sent, hipEventDisableTiming));
    }
    c->attached = true;
    return 0;
}

// "Every reader of the receive slab's current contents has been enqueued on `stream`": call it BEFORE enqueueing the forward whose
// states the next S3ENC_EXCHANGE_COPY exchange moves — that exchange then tells the peers "you may write" at this point of the stream
// instead of at the point of its own call, i.e. the pushes behind the layer events overlap the forward.  Optional (see copy_exchange).
int s3enc_comm_copy_release(s3enc_comm c, void* stream) {
    if (!c) return fail("s3enc_comm_copy_release: null argument");
    DeviceGuard dg(c->device);
    if (!c->free_ev) HIP_TRY(hipEventCreateWithFlags(&c->free_ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(c->free_ev, (hipStream_t)stream));
    c->have_free = true;
    return 0;
}

// 0 = every wait of every S3ENC_EXCHANGE_COPY exchange so far met its deadline; else bit p = a wait for rank p (mod 32) timed out
// (S3ENC_COPY_DEADLINE_MS, default 5000) — the slab's contents are then not valid.  Synchronises the communicator's streams.
int s3enc_comm_copy_status(s3enc_comm c, int32_t* status) {
    if (!c || !status) return fail("s3enc_comm_copy_status: null argument");
    *status = 0;
    if (!c->mbox) return 0;
    DeviceGuard dg(c->device);
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (CopyPeer& pr : c->peers)
        if (pr.stream) HIP_TRY(hipStreamSynchronize(pr.stream));
    uint64_t e = 0;
    HIP_TRY(hipMemcpy(&e, c->mbox + 2 * c->world, sizeof(e), hipMemcpyDeviceToHost));
    *status = (int32_t)(e & 0x7fffffff);
    return 0;
}

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
    for (CopyPeer& pr : c->peers) {
        if (pr.stream) (void)hipStreamSynchronize(pr.stream);
        if (pr.sent) (void)hipEventDestroy(pr.sent);
        if (pr.stream) (void)hipStreamDestroy(pr.stream);
        if (pr.slab_map) (void)hipIpcCloseMemHandle(pr.slab_map);
        if (pr.mbox_map) (void)hipIpcCloseMemHandle(pr.mbox_map);
    }
    if (c->mbox) (void)hipFree(c->mbox);
    if (c->free_ev) (void)hipEventDestroy(c->free_ev);
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
    if (!c || !send || !recv || n_states < 1 || bytes_per_state < 1) return fail("s3enc_comm_exchange_states: bad arguments");
    if (algo == S3ENC_EXCHANGE_COPY)
        return copy_exchange(c, send, send_state_stride, recv, recv_state_stride, n_states, bytes_per_state, ready_events, stream);
    Rccl& R = rccl();
    if (!R.lib || !R.why.empty()) return fail("s3enc_comm: " + R.why);
    if (algo != S3ENC_EXCHANGE_COLLECTIVE && algo != S3ENC_EXCHANGE_DIRECT) return fail("s3enc_comm_exchange_states: unknown algo");
    if (!c || !c->comm) return fail("s3enc_comm_exchange_states: this communicator has no RCCL side (s3enc_comm_init_local): S3ENC_EXCHANGE_COPY only");
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
