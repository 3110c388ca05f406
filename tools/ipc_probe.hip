// ipc_probe.hip -- what the sharded particle filter's device-side exchange relies on, probed with TWO PROCESSES ON ONE CARD
// (a one-GPU box is all the builder has):
//   1. hipIpcGetMemHandle / hipIpcOpenMemHandle of (a) plain hipMalloc memory, (b) fine-grained and (c) uncached device memory
//   2. kernels of the two processes running CONCURRENTLY and handing a sequence number back and forth through the
//      peer-mapped buffer (system-scope relaxed stores / loads, bounded spin): round trips per second
//   3. bulk read of the peer's buffer after the hand-shake (system-scope loads), checked
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/ipc_probe tools/ipc_probe.hip ; run: tools/ipc_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>

#define CK(e)                                                                                  \
    do {                                                                                       \
        hipError_t _e = (e);                                                                   \
        if (_e != hipSuccess) {                                                                \
            fprintf(stderr, "[rank %d] %s: %s (line %d)\n", g_rank, #e, hipGetErrorString(_e), __LINE__); \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)

static int g_rank = -1;

__global__ void pingpong(unsigned long long* mine, unsigned long long* peer, int rank, int iters, unsigned long long* out) {
    // rank 0 sends 1, waits for 1 from the peer, sends 2, ...   the peer echoes
    const unsigned long long t0 = wall_clock64();
    int done = 0;
    for (int i = 1; i <= iters; ++i) {
        if (rank == 0) __hip_atomic_store(peer, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        bool ok = true;
        while (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned long long)i) {
            __builtin_amdgcn_s_sleep(4);
            if (wall_clock64() - t0 > 500000000ull) { ok = false; break; }       // 5 s
        }
        if (!ok) break;
        if (rank == 1) __hip_atomic_store(peer, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        done = i;
    }
    out[0] = done;
    out[1] = wall_clock64() - t0;
}

__global__ void fill(unsigned* buf, int n, unsigned v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) buf[i] = v + i;
}

// signal "my fill kernel is complete" (stream order) and wait for the peer's: ONE workgroup.  (First form of this probe: every
// workgroup of the 4096-workgroup check kernel polled -- on a SHARED card the spinning grid fills every CU and the peer's
// kernel, which would have to send the flag, never starts: time-outs.  A hand-shake between ranks that may share a card
// belongs in a one-workgroup kernel.)
__global__ void gate(unsigned long long* mine, unsigned long long* peer, unsigned long long tag, unsigned* bad) {
    __hip_atomic_store(peer, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != tag) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > 500000000ull) { atomicAdd(bad, 1000000u); break; }
    }
}

// read the peer's payload with system-scope loads
__global__ void check(const unsigned* peer_payload, int n, unsigned v, unsigned* bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const unsigned got = __hip_atomic_load(peer_payload + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (got != v + i) atomicAdd(bad, 1u);
    }
}

static int run(int rank, int rfd, int wfd) {
    g_rank = rank;
    setvbuf(stdout, nullptr, _IONBF, 0);
    CK(hipSetDevice(0));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const char* kinds[3] = {"hipMalloc", "fine-grained", "uncached"};
    for (int kind = 0; kind < 3; ++kind) {
        const int N = 1 << 20;
        unsigned long long* flag = nullptr;        // [0] ping-pong word, [8] hand-shake word (separate lines)
        unsigned* payload = nullptr;
        hipError_t e;
        if (kind == 0) {
            e = hipMalloc((void**)&flag, 4096);
            if (e == hipSuccess) e = hipMalloc((void**)&payload, sizeof(unsigned) * N);
        } else {
            const unsigned fl = kind == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached;
            e = hipExtMallocWithFlags((void**)&flag, 4096, fl);
            if (e == hipSuccess) e = hipExtMallocWithFlags((void**)&payload, sizeof(unsigned) * N, fl);
        }
        int ok_local = e == hipSuccess;
        hipIpcMemHandle_t hs[2];
        memset(hs, 0, sizeof(hs));
        if (ok_local) {
            CK(hipMemset(flag, 0, 4096));
            e = hipIpcGetMemHandle(&hs[0], flag);
            if (e == hipSuccess) e = hipIpcGetMemHandle(&hs[1], payload);
            ok_local = e == hipSuccess;
        }
        if (!ok_local) fprintf(stderr, "[rank %d] %s: alloc/export failed: %s\n", rank, kinds[kind], hipGetErrorString(e));
        char msg[1 + sizeof(hs)];
        msg[0] = (char)ok_local;
        memcpy(msg + 1, hs, sizeof(hs));
        if (write(wfd, msg, sizeof(msg)) != (ssize_t)sizeof(msg)) return 1;
        char in[1 + sizeof(hs)];
        size_t got = 0;
        while (got < sizeof(in)) {
            const ssize_t r = read(rfd, in + got, sizeof(in) - got);
            if (r <= 0) return 1;
            got += (size_t)r;
        }
        if (!ok_local || !in[0]) { printf("[rank %d] %-12s SKIPPED (export failed on a rank)\n", rank, kinds[kind]); continue; }
        hipIpcMemHandle_t ph[2];
        memcpy(ph, in + 1, sizeof(ph));
        unsigned long long* pflag = nullptr;
        unsigned* ppay = nullptr;
        e = hipIpcOpenMemHandle((void**)&pflag, ph[0], hipIpcMemLazyEnablePeerAccess);
        if (e == hipSuccess) e = hipIpcOpenMemHandle((void**)&ppay, ph[1], hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) { printf("[rank %d] %-12s open failed: %s\n", rank, kinds[kind], hipGetErrorString(e)); continue; }
        printf("[rank %d] %-12s flag %p payload %p | peer flag %p payload %p\n", rank, kinds[kind], (void*)flag, (void*)payload,
               (void*)pflag, (void*)ppay);
        unsigned long long* d_out;
        unsigned* d_bad;
        CK(hipMalloc((void**)&d_out, 16));
        CK(hipMalloc((void**)&d_bad, 4));
        CK(hipMemset(d_bad, 0, 4));
        const int iters = 2000;
        hipLaunchKernelGGL(pingpong, dim3(1), dim3(1), 0, s, flag, pflag, rank, iters, d_out);
        CK(hipGetLastError());
        CK(hipStreamSynchronize(s));
        unsigned long long out[2];
        CK(hipMemcpy(out, d_out, 16, hipMemcpyDeviceToHost));
        printf("[rank %d] %-12s ping-pong: %llu of %d round trips, %.2f us each\n", rank, kinds[kind], out[0], iters,
               out[0] ? (double)out[1] * 0.01 / (double)out[0] : 0.0);
        // payload + hand-shake, three rounds with different contents (a stale cached line of the peer's payload would show)
        unsigned bad_total = 0;
        for (int round = 1; round <= 3; ++round) {
            const unsigned v = 0x1000000u * (unsigned)(round + 4 * rank);
            const unsigned pv = 0x1000000u * (unsigned)(round + 4 * (1 - rank));
            printf("[rank %d] %-12s round %d\n", rank, kinds[kind], round);
            hipLaunchKernelGGL(fill, dim3(N / 256), dim3(256), 0, s, payload, N, v);
            hipLaunchKernelGGL(gate, dim3(1), dim3(1), 0, s, flag + 8, pflag + 8, (unsigned long long)(100 * (kind + 1) + round), d_bad);
            hipLaunchKernelGGL(check, dim3(N / 256), dim3(256), 0, s, (const unsigned*)ppay, N, pv, d_bad);
            CK(hipGetLastError());
            CK(hipStreamSynchronize(s));
            unsigned bad;
            CK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
            bad_total += bad;
            // nobody may overwrite its payload before the peer has read it: a host-level barrier through the pipes
            char c = 1;
            if (write(wfd, &c, 1) != 1 || read(rfd, &c, 1) != 1) return 1;
        }
        printf("[rank %d] %-12s payload after hand-shake: %u wrong words of %d x 3\n", rank, kinds[kind], bad_total, N);
        CK(hipIpcCloseMemHandle(pflag));
        CK(hipIpcCloseMemHandle(ppay));
        CK(hipDeviceSynchronize());
        char c = 1;
        if (write(wfd, &c, 1) != 1 || read(rfd, &c, 1) != 1) return 1;      // both have closed before anyone frees
        CK(hipFree(flag));
        CK(hipFree(payload));
        CK(hipFree(d_out));
        CK(hipFree(d_bad));
    }
    return 0;
}

int main() {
    int a[2], b[2];
    if (pipe(a) || pipe(b)) return 1;
    const pid_t pid = fork();                 // before anything touches the GPU
    if (pid == 0) return run(1, a[0], b[1]);
    const int rc = run(0, b[0], a[1]);
    int st = 0;
    waitpid(pid, &st, 0);
    printf("probe exit: rank0 %d rank1 %d\n", rc, WEXITSTATUS(st));
    return rc || WEXITSTATUS(st);
}
