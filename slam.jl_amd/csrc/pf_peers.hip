// pf_peers.hip -- the sharding of the particle filter behind the C ABI (SURVEY 8b `n_devices`, 8e): one process per GPU,
// IPC-mapped peer buffers, hand-shakes through the inboxes.  The library links no collective library.
// Reference: none (the particle types of src/common.jl:14-20,31-34 say nothing about devices).
#include "pf_device.h"

namespace {

//             a rank publishes only after its own resampling s has completed (stream order).
// one lane: "rank `rank` is going away" into every peer's inbox (slam_pf_destroy of a handle that is still attached)
__global__ void pf_peer_gone_kernel(const PfPeers* __restrict__ peers, int rank, int world) {
    const int r = threadIdx.x;
    if (r < world && r != rank) __hip_atomic_store(&peers->inbox[r]->gone[rank][0], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(64) void pf_peer_gate_kernel(PfCtl* ctl, long long seq, const PfPeers* __restrict__ peers,
                                                          PfInbox* inbox, int rank, int world) {
    if (ctl->resample_seq != seq || ctl->error != 0) return;
    if (pf_peer_gone(inbox, world)) { ctl->error = PF_ERR_PEER; return; }       // (the kernels behind this one return on ctl->error)
    const int r = threadIdx.x;
    if (r < world) {
        __hip_atomic_store(&peers->inbox[r]->ready[rank][0], (unsigned long long)seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(&inbox->ready[r][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned long long)seq) {
            __builtin_amdgcn_s_sleep(20);
            if (wall_clock64() - t0 > 2000000000ull) { ctl->error = PF_ERR_PEER; break; }      // 20 s: a rank is gone
        }
    }
}

// A barrier among the ranks on their streams (materialise): every rank counts its calls, tells every peer, waits for all.
__global__ __launch_bounds__(64) void pf_peer_barrier_kernel(int32_t* err, unsigned long long count, const PfPeers* __restrict__ peers,
                                                             PfInbox* inbox, int rank, int world, unsigned long long timeout_ticks) {
    if (pf_peer_gone(inbox, world)) { *err = PF_ERR_PEER; return; }
    const int r = threadIdx.x;
    if (r < world) {
        __hip_atomic_store(&peers->inbox[r]->bar[rank][0], count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(&inbox->bar[r][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < count) {
            __builtin_amdgcn_s_sleep(20);
            if (wall_clock64() - t0 > timeout_ticks) { *err = PF_ERR_PEER; break; }
        }
    }
}

}  // namespace

int pf_launch_peer_barrier(slam_pf* h, int32_t* d_err, unsigned long long count, unsigned long long timeout_ticks) {
    hipLaunchKernelGGL(pf_peer_barrier_kernel, dim3(1), dim3(64), 0, h->stream, d_err, count, (const PfPeers*)h->d_peers, h->inbox,
                       h->xchg_rank, h->xchg_world, timeout_ticks);
    HIP_TRY(hipGetLastError());
    return SLAM_OK;
}

int pf_peer_barrier(slam_pf* h) {
    h->bar_count += 1;
    return pf_launch_peer_barrier(h, &h->d_ctl->error, (unsigned long long)h->bar_count, 2000000000ull);       // 20 s
}

int pf_launch_peer_gate(slam_pf* h, long long seq) {
    hipLaunchKernelGGL(pf_peer_gate_kernel, dim3(1), dim3(64), 0, h->stream, h->d_ctl, seq, (const PfPeers*)h->d_peers, h->inbox,
                       h->xchg_rank, h->xchg_world);
    HIP_TRY(hipGetLastError());
    return SLAM_OK;
}

void pf_announce_gone(slam_pf* h) {
    hipLaunchKernelGGL(pf_peer_gone_kernel, dim3(1), dim3(64), 0, h->stream, (const PfPeers*)h->d_peers, h->xchg_rank, h->xchg_world);
    (void)hipGetLastError();
    (void)hipStreamSynchronize(h->stream);
}

/* ---- sharding behind the C ABI: peers -----------------------------------------------------------------------------
 * One process per GPU.  Every rank exports a blob (slam_pf_export_peer: IPC handles of its state buffers and of its
 * inbox page, or -- same process, e.g. one host thread per GPU -- the raw device pointers), the caller moves the blobs
 * between the ranks by whatever it has (MPI, files, torch.distributed ...), and every rank attaches all of them in
 * rank order.  From then on slam_pf_step_auto resamples the sharded filter on the device (no SLAM_PF_HALTED). */
constexpr int PF_BLOB_FIXED = 7;                                  // pose0, pose1, logw0, logw1, tab0, tab1, inbox
constexpr int PF_BLOB_MAXH = PF_BLOB_FIXED + 2 * PF_LM_MAXC;      // ... then the landmark chunks: buffer 0's, buffer 1's
struct PfPeerBlob {
    uint64_t magic;
    int64_t pid;
    int32_t device, dtype, nl, lm_shift, lm_nchunks, reserved;
    int64_t n, n_global;
    uint64_t lm_chunk_bytes, inbox_bytes;
    void* raw[PF_BLOB_MAXH];
    hipIpcMemHandle_t ipc[PF_BLOB_MAXH];
};
static_assert(sizeof(PfPeerBlob) <= SLAM_PF_PEER_BLOB_BYTES, "peer blob");
constexpr uint64_t PF_BLOB_MAGIC = 0x534c414d50465034ull;      // "SLAMPFP4"
// What an IPC mapping may carry on this runtime (ROCm 7.2, dmabuf IPC; DESIGN section 7 has the records):
//   * a sharded filter whose exported buffers exceeded 2 GiB HUNG in its attach flow (round 3: 1.91 GiB attached in
//     milliseconds, 2.50 GiB left both processes waiting, tools/ipc_gen_test.py big:*).  The cause is NOT located: opening a
//     2.5 GiB allocation by itself, alone and pairwise, returns at once (profiles/r04_ipc_open_stack.txt, r04_ipc_pair_probe.txt),
//     so it is something in the filter's own attach or first step above the 2^31-byte boundary, not the runtime call.  Until
//     it is found every exported buffer stays below PF_IPC_MAX_BYTES -- the landmark records are chunked for that reason;
//   * the import of a FINE-GRAINED (hipExtMallocWithFlags) allocation larger than one 2 MiB fragment was seen with only its
//     first 2 MiB mapped (tools/ipc_probe.hip: page fault at import + 2 MiB in 3 of 7 runs; plain hipMalloc imports of the same
//     size never): the only fine-grained export is the inbox, which must stay within PF_IPC_FINE_MAX_BYTES.
// slam_pf_attach_peers checks both BEFORE opening anything and refuses with SLAM_E_CAPACITY (the caller keeps the halting flow).
constexpr uint64_t PF_IPC_MAX_BYTES = 2047ull << 20;
constexpr uint64_t PF_IPC_FINE_MAX_BYTES = 2ull << 20;

static int pf_blob_handles(const slam_pf* h, void* ptrs[PF_BLOB_MAXH]) {
    ptrs[0] = h->pose[0]; ptrs[1] = h->pose[1]; ptrs[2] = h->logw2[0]; ptrs[3] = h->logw2[1];
    ptrs[4] = h->d_tab[0]; ptrs[5] = h->d_tab[1]; ptrs[6] = h->inbox;
    int cnt = PF_BLOB_FIXED;
    for (int b = 0; b < 2; ++b)
        for (int k = 0; k < h->lmtab.nchunks; ++k) ptrs[cnt++] = h->lmtab.c[b][k];
    return cnt;
}

extern "C" int slam_pf_export_peer(slam_pf_t h, void* blob) {
    ARG_CHECK(h != nullptr && blob != nullptr, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    PfPeerBlob b;
    memset(&b, 0, sizeof(b));
    b.magic = PF_BLOB_MAGIC;
    b.pid = (int64_t)getpid();
    b.device = h->device; b.dtype = h->dtype; b.nl = h->nl; b.n = h->n; b.n_global = h->n_global;
    b.lm_shift = h->lmtab.shift; b.lm_nchunks = h->lmtab.nchunks; b.lm_chunk_bytes = h->lm_chunk_bytes; b.inbox_bytes = h->inbox_bytes;
    void* ptrs[PF_BLOB_MAXH];
    const int cnt = pf_blob_handles(h, ptrs);
    for (int i = 0; i < cnt; ++i) {
        b.raw[i] = ptrs[i];
        HIP_TRY(hipIpcGetMemHandle(&b.ipc[i], ptrs[i]));
    }
    memset(blob, 0, SLAM_PF_PEER_BLOB_BYTES);
    memcpy(blob, &b, sizeof(b));
    return SLAM_OK;
}

void pf_detach_peers_impl(slam_pf* h) {
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (int r = 0; r < PF_MAX_WORLD; ++r)
        for (int i = 0; i < PF_BLOB_MAXH; ++i)
            if (h->peer_open[r][i]) {
                (void)hipIpcCloseMemHandle(h->peer_open[r][i]);
                h->peer_open[r][i] = nullptr;
            }
    (void)hipDeviceSynchronize();       // the unmaps have taken effect before anybody frees (and re-exports) the memory behind them
    if (h->d_peers) { (void)hipFree(h->d_peers); h->d_peers = nullptr; }
    memset(&h->peers, 0, sizeof(h->peers));
}

extern "C" int slam_pf_detach_peers(slam_pf_t h) {
    SLAM_RANGE();
    ARG_CHECK(h != nullptr, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    PF_LEGACY_ENTRY(h);
    { const int rcm = pf_materialise(h); if (rcm) return rcm; }      // (collective: no remote references may stay behind)
    pf_detach_peers_impl(h);
    if (!h->xchg_host) { h->xchg_rank = 0; h->xchg_world = 1; }
    return SLAM_OK;
}

extern "C" int slam_pf_attach_peers(slam_pf_t h, int rank, int world, const void* blobs) {
    SLAM_RANGE();
    ARG_CHECK(h != nullptr && blobs != nullptr, "null argument");
    ARG_CHECK(world >= 1 && world <= PF_MAX_WORLD && rank >= 0 && rank < world, "rank / world out of range (at most 8 ranks)");
    ARG_CHECK(h->n * world == h->n_global && h->first == (int64_t)rank * h->n, "ranks must own equal, contiguous slices in rank order");
    HIP_TRY(hipSetDevice(h->device));
    PF_LEGACY_ENTRY(h);
    ARG_CHECK(!pf_sharded(h), "peers are already attached (slam_pf_detach_peers first)");
    pf_detach_peers_impl(h);
    const int64_t me = (int64_t)getpid();
    PfPeers t;
    memset(&t, 0, sizeof(t));
    // pass 0 checks every blob (nothing is opened before all of them are acceptable), pass 1 opens
    for (int pass = 0; pass < 2; ++pass)
    for (int r = 0; r < world; ++r) {
        PfPeerBlob b;
        memcpy(&b, (const char*)blobs + (size_t)r * SLAM_PF_PEER_BLOB_BYTES, sizeof(b));
        const int cnt = PF_BLOB_FIXED + 2 * b.lm_nchunks;
        if (pass == 0) {
            ARG_CHECK(b.magic == PF_BLOB_MAGIC, "a peer blob is not one of slam_pf_export_peer's");
            ARG_CHECK(b.n == h->n && b.nl == h->nl && b.dtype == h->dtype && b.n_global == h->n_global, "the peers' shards differ in size or type");
            ARG_CHECK(b.lm_shift == h->lmtab.shift && b.lm_nchunks == h->lmtab.nchunks && b.lm_nchunks >= 1 && b.lm_nchunks <= PF_LM_MAXC,
                      "the peers' landmark chunking differs");
            if (r == rank) ARG_CHECK(b.pid == me && b.raw[0] == h->pose[0], "blob [rank] is not this handle's own export");
            if (r != rank && b.pid != me) {
                // the shapes this runtime's IPC mappings cannot carry (see PF_IPC_MAX_BYTES): refuse BEFORE opening anything
                const uint64_t n64 = (uint64_t)b.n, esz = (uint64_t)h->esz;
                const uint64_t largest = std::max<uint64_t>(std::max<uint64_t>(3 * n64 * esz, (uint64_t)PF_TAB_MAX * n64 * 4), b.lm_chunk_bytes);
                if (largest > PF_IPC_MAX_BYTES) {
                    slam_set_error("rank %d exports a buffer of %.2f GiB: a filter with exported buffers above 2 GiB hung in its attach flow "
                                   "(cause unknown) and is refused (use more ranks, or the halting flow)", r, (double)largest / 1073741824.0);
                    return SLAM_E_CAPACITY;
                }
                if (b.inbox_bytes > PF_IPC_FINE_MAX_BYTES) {
                    slam_set_error("rank %d's inbox is %.2f MiB: a fine-grained allocation above 2 MiB is not exported (it was seen "
                                   "half mapped on this runtime); use the halting flow for a filter of this size", r,
                                   (double)b.inbox_bytes / 1048576.0);
                    return SLAM_E_CAPACITY;
                }
            }
            continue;
        }
        void* ptr[PF_BLOB_MAXH];
        if (r == rank || b.pid == me) {                     // this handle, or a shard of this very process: plain pointers
            if (r != rank && b.device != h->device) {
                const hipError_t e = hipDeviceEnablePeerAccess(b.device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIP_TRY(e);
                (void)hipGetLastError();
            }
            for (int i = 0; i < cnt; ++i) ptr[i] = b.raw[i];
        } else {
            for (int i = 0; i < cnt; ++i) {
                const hipError_t e = hipIpcOpenMemHandle(&ptr[i], b.ipc[i], hipIpcMemLazyEnablePeerAccess);
                if (e != hipSuccess) {
                    slam_set_error("hipIpcOpenMemHandle of rank %d's buffer %d failed: %s", r, i, hipGetErrorString(e));
                    pf_detach_peers_impl(h);
                    return SLAM_E_HIP;
                }
                h->peer_open[r][i] = ptr[i];
            }
        }
        t.pose[r][0] = ptr[0]; t.pose[r][1] = ptr[1]; t.logw[r][0] = ptr[2]; t.logw[r][1] = ptr[3];
        t.tab[r][0] = (int32_t*)ptr[4]; t.tab[r][1] = (int32_t*)ptr[5];
        t.inbox[r] = (PfInbox*)ptr[6];
        t.lm[r].shift = b.lm_shift; t.lm[r].nchunks = b.lm_nchunks;
        for (int bb = 0; bb < 2; ++bb)
            for (int k = 0; k < b.lm_nchunks; ++k) t.lm[r].c[bb][k] = ptr[PF_BLOB_FIXED + bb * b.lm_nchunks + k];
    }
    h->peers = t;
    HIP_TRY(hipMalloc((void**)&h->d_peers, sizeof(PfPeers)));
    HIP_TRY(hipMemcpy(h->d_peers, &t, sizeof(t), hipMemcpyHostToDevice));
    h->xchg_rank = rank;
    h->xchg_world = world;
    return SLAM_OK;
}

/* A barrier among the attached ranks through their inboxes (every rank writes a word into every peer's inbox and polls its
 * own): collective, synchronises.  SLAM_OK when every peer's word arrived within timeout_ms -- the caller's check that the
 * GPUs really see each other's writes before it relies on the device-side exchange (it can fall back to the halting
 * flow otherwise). */
extern "C" int slam_pf_peer_selftest(slam_pf_t h, int timeout_ms) {
    ARG_CHECK(h != nullptr, "null handle");
    ARG_CHECK(pf_sharded(h), "no peers attached");
    ARG_CHECK(timeout_ms > 0, "timeout must be positive");
    HIP_TRY(hipSetDevice(h->device));
    PF_LEGACY_ENTRY(h);
    int32_t* d_err = nullptr;
    HIP_TRY(hipMalloc((void**)&d_err, sizeof(int32_t)));
    hipError_t e = hipMemsetAsync(d_err, 0, sizeof(int32_t), h->stream);
    h->bar_count += 1;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(pf_peer_barrier_kernel, dim3(1), dim3(64), 0, h->stream, d_err, (unsigned long long)h->bar_count,
                           (const PfPeers*)h->d_peers, h->inbox, h->xchg_rank, h->xchg_world, (unsigned long long)timeout_ms * 100000ull);
        e = hipGetLastError();
    }
    int32_t err = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&err, d_err, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(d_err);
    if (e != hipSuccess) {
        slam_set_error("HIP error in slam_pf_peer_selftest: %s", hipGetErrorString(e));
        return SLAM_E_HIP;
    }
    if (err) {
        slam_set_error("peer self-test: a peer's word did not arrive within %d ms", timeout_ms);
        return SLAM_E_HIP;
    }
    return SLAM_OK;
}

/* out = {ranks of the filter, 1 if peers are attached (device-side resampling of the sharded filter), SLAM_PF_HALTED
 * returns so far, resamplings so far}. */
extern "C" int slam_pf_comm_info(slam_pf_t h, int64_t out[4]) {
    ARG_CHECK(h != nullptr && out != nullptr, "null argument");
    out[0] = h->xchg_world;
    out[1] = pf_sharded(h) ? 1 : 0;
    out[2] = h->halts;
    int64_t cnt = 0;
    const int rc = slam_pf_resample_count(h, &cnt);
    if (rc) return rc;
    out[3] = cnt;
    return SLAM_OK;
}
