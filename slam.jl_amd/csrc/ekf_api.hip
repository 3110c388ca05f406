// ekf_api.hip -- the extern "C" surface of libslamhip.so for the EKF path
// (declared in include/slamhip.h).  Host-side bookkeeping only; every numeric
// operation on the state is a HIP kernel in the other translation units.  There
// is deliberately no CPU fallback: without a usable device create() fails.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "common.h"

int update_kernels_init();   // ekf_update.hip

// ---- errors ------------------------------------------------------------------
static thread_local char g_err[512] = "";

void slam_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* slam_last_error(void) { return g_err; }

// ---- roctx (see common.h) ----------------------------------------------------------
#include <dlfcn.h>
#include <mutex>
namespace {
typedef int (*roctx_push_fn)(const char*);
typedef int (*roctx_pop_fn)(void);
roctx_push_fn g_roctx_push = nullptr;
roctx_pop_fn g_roctx_pop = nullptr;
std::once_flag g_roctx_once;
void roctx_resolve() {
    const char* names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"};
    const char* env = getenv("SLAMHIP_ROCTX");
    const bool want = env && atoi(env) != 0;
    if (env && !want) return;                                     // SLAMHIP_ROCTX=0: never
    void* lib = nullptr;
    for (const char* n : names)
        if ((lib = dlopen(n, RTLD_LAZY | RTLD_NOLOAD))) break;      // the profiler (or the application) has it loaded already
    if (!lib && want)
        for (const char* n : names)
            if ((lib = dlopen(n, RTLD_LAZY | RTLD_GLOBAL))) break;
    if (!lib) return;
    g_roctx_push = (roctx_push_fn)dlsym(lib, "roctxRangePushA");
    g_roctx_pop = (roctx_pop_fn)dlsym(lib, "roctxRangePop");
    if (!g_roctx_push || !g_roctx_pop) g_roctx_push = nullptr, g_roctx_pop = nullptr;
}
}  // namespace
void slam_roctx_push(const char* name) {
    std::call_once(g_roctx_once, roctx_resolve);
    if (g_roctx_push) (void)g_roctx_push(name);
}
void slam_roctx_pop(void) {
    if (g_roctx_pop) (void)g_roctx_pop();
}

extern "C" int slam_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

// ---- timing ------------------------------------------------------------------
KTimer::KTimer(slam_ekf* h_, int kid, hipStream_t stream)
    : h(h_), on(h_->timing == 1 || ((h_->timing >> (kid + 1)) & 1)), s(stream ? stream : h_->stream) {
    if (!on) return;
    if (!h->free_pairs.empty()) {
        p = h->free_pairs.back();
        h->free_pairs.pop_back();
    } else {
        // timing only: no system-scope fence (cache write-back + invalidate) when the event fires
        if (hipEventCreateWithFlags(&p.a, hipEventDisableSystemFence) != hipSuccess ||
            hipEventCreateWithFlags(&p.b, hipEventDisableSystemFence) != hipSuccess) {
            on = false;
            return;
        }
    }
    p.kid = kid;
    (void)hipEventRecord(p.a, s);
}

KTimer::~KTimer() {
    if (!on) return;
    (void)hipEventRecord(p.b, s);
    h->pairs.push_back(p);
}

static int fold_timing(slam_ekf* h) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (auto& p : h->pairs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            h->t_ms[p.kid] += ms;
            h->t_sq[p.kid] += (double)ms * (double)ms;
            if (h->t_min[p.kid] == 0.0 || ms < h->t_min[p.kid]) h->t_min[p.kid] = ms;
            h->t_n[p.kid] += 1;
        }
        h->free_pairs.push_back(p);
    }
    h->pairs.clear();
    return SLAM_OK;
}

// ---- allocation helpers --------------------------------------------------------
template <typename P>
static int dev_alloc_zero(P** p, size_t bytes, hipStream_t s) {
    *p = nullptr;
    if (bytes == 0) bytes = 16;
    HIP_TRY(hipMalloc((void**)p, bytes));
    HIP_TRY(hipMemsetAsync(*p, 0, bytes, s));
    return SLAM_OK;
}

static void dev_free(void* p) {
    if (p) (void)hipFree(p);
}

int ensure_obs_capacity(slam_ekf* h, int nobs) {
    if (nobs <= h->ocap) return SLAM_OK;
    int cap = h->ocap ? h->ocap : 256;
    while (cap < nobs) cap *= 2;
    HIP_TRY(hipStreamSynchronize(h->stream));
    dev_free(h->obsbuf); dev_free(h->idfbuf); dev_free(h->d_assoc); dev_free(h->gate_part); dev_free(h->znbuf);
    if (h->h_obs) (void)hipHostFree(h->h_obs);
    if (h->h_idf) (void)hipHostFree(h->h_idf);
    if (h->h_assoc) (void)hipHostFree(h->h_assoc);
    h->obsbuf = nullptr; h->idfbuf = nullptr; h->d_assoc = nullptr; h->gate_part = nullptr; h->znbuf = nullptr;
    h->h_obs = nullptr; h->h_idf = nullptr; h->h_assoc = nullptr;
    h->ocap = 0;
    int rc;
    if ((rc = dev_alloc_zero(&h->obsbuf, sizeof(double) * 2 * cap, h->stream))) return rc;
    if ((rc = dev_alloc_zero(&h->idfbuf, sizeof(int32_t) * cap, h->stream))) return rc;
    if ((rc = dev_alloc_zero(&h->d_assoc, sizeof(int32_t) * cap, h->stream))) return rc;
    if ((rc = dev_alloc_zero(&h->znbuf, sizeof(double) * 2 * cap, h->stream))) return rc;
    // gating partials: [blocks][<=256 observations per sweep][3]
    h->gate_blocks_cap = (h->maxN + 63) / 64 + 1;
    if ((rc = dev_alloc_zero(&h->gate_part, sizeof(double) * 3 * 256 * (size_t)h->gate_blocks_cap, h->stream))) return rc;
    HIP_TRY(hipHostMalloc((void**)&h->h_obs, sizeof(double) * 2 * cap, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void**)&h->h_idf, sizeof(int32_t) * cap, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void**)&h->h_assoc, sizeof(int32_t) * cap, hipHostMallocDefault));
    HIP_TRY(hipHostGetDevicePointer((void**)&h->h_obs_dev, h->h_obs, 0));
    HIP_TRY(hipHostGetDevicePointer((void**)&h->h_assoc_dev, h->h_assoc, 0));
    h->ocap = cap;
    return SLAM_OK;
}

static void free_update_workspace(slam_ekf* h) {
    dev_free(h->PHt); dev_free(h->PHtS); dev_free(h->Kd); dev_free(h->W1); dev_free(h->W2); dev_free(h->Cmat); dev_free(h->Wimg);
    dev_free(h->Smat); dev_free(h->Mwork); dev_free(h->gvec);
    h->W1 = h->W2 = h->Wimg = nullptr;
    h->PHt = h->PHtS = h->Kd = h->Cmat = h->Smat = h->Mwork = h->gvec = nullptr;
    h->kcap = 0;
}

static int zero_panels(slam_ekf* h) {
    if (!h->kcap) return SLAM_OK;
    HIP_TRY(hipMemsetAsync(h->PHt, 0, sizeof(double) * (size_t)h->npad * h->kcap, h->stream));
    HIP_TRY(hipMemsetAsync(h->Kd, 0, sizeof(double) * (size_t)h->npad * h->kcap, h->stream));
    HIP_TRY(hipMemsetAsync(h->W1, 0, h->esz * (size_t)h->npad * 2 * h->kcap, h->stream));
    HIP_TRY(hipMemsetAsync(h->W2, 0, h->esz * (size_t)h->npad * 2 * h->kcap, h->stream));
    return SLAM_OK;
}

int ensure_update_workspace(slam_ekf* h, int m) {
    const int kp = round_up(2 * m, SLAM_KPAD);
    if (kp <= h->kcap) return SLAM_OK;
    int cap = h->kcap ? h->kcap : 32;
    while (cap < kp) cap *= 2;
    HIP_TRY(hipStreamSynchronize(h->stream));
    free_update_workspace(h);
    int rc;
    if ((rc = dev_alloc_zero(&h->PHt, sizeof(double) * (size_t)h->npad * cap, h->stream))) return rc;
    if ((rc = dev_alloc_zero(&h->PHtS, sizeof(double) * (size_t)(3 + cap) * cap, h->stream))) return rc;
    if ((rc = dev_alloc_zero(&h->Kd, sizeof(double) * (size_t)h->npad * cap, h->stream))) return rc;
    if ((rc = dev_alloc_zero(&h->W1, h->esz * (size_t)h->npad * 2 * cap, h->stream))) return rc;
    if ((rc = dev_alloc_zero(&h->W2, h->esz * (size_t)h->npad * 2 * cap, h->stream))) return rc;
    if (h->dtype == SLAM_F32 && (rc = dev_alloc_zero(&h->Wimg, (size_t)(h->npad / 128) * (cap / 16) * 3 * 4096, h->stream))) return rc;
    if ((rc = dev_alloc_zero(&h->Cmat, sizeof(double) * (size_t)cap * cap, h->stream))) return rc;
    if ((rc = dev_alloc_zero(&h->Smat, sizeof(double) * (size_t)cap * cap, h->stream))) return rc;
    if ((rc = dev_alloc_zero(&h->Mwork, sizeof(double) * (size_t)cap * (cap + 1), h->stream))) return rc;
    if ((rc = dev_alloc_zero(&h->gvec, sizeof(double) * (size_t)cap, h->stream))) return rc;
    h->kcap = cap;
    return SLAM_OK;
}

// ---- create / destroy -----------------------------------------------------------
extern "C" int slam_ekf_destroy(slam_ekf_t h) {
    if (!h) return SLAM_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
#ifdef SLAMHIP_EXPERIMENTS
    if (h->dd_prof && slam_exp_env("SLAMHIP_STAMPS", 0)) {       // dd_stream_dma's phase sums of the LAST down-date
        const size_t cnt = (size_t)4096 * 16;
        std::vector<unsigned long long> v(cnt);
        if (hipMemcpy(v.data(), h->dd_prof, cnt * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            double s[16] = {0};
            size_t waves = 0;
            for (size_t w = 0; w < 4096; ++w) {
                if (!v[16 * w]) continue;
                ++waves;
                for (int i = 0; i < 16; ++i) s[i] += (double)v[16 * w + i];
            }
            if (waves) {
                fprintf(stderr, "[slamhip] dma down-date, mean per wave of %zu waves: steps %.1f  clk per step: reads %.0f  mfma-issue %.0f  barrier %.0f  dma-issue %.0f  stores(per tile) %.0f  chunk-wait by step:",
                        waves, s[0] / waves, s[1] / s[0], s[2] / s[0], s[3] / s[0], s[4] / s[0], s[5] / s[0] * 8);
                for (int i = 0; i < 8; ++i) fprintf(stderr, " %.0f", s[6 + i] / s[0] * 8);
                fprintf(stderr, "  | stream: %.0f shader clocks in %.2f us per wave = %.3f GHz in-kernel clock\n", s[14] / waves, s[15] / waves / 100.0,
                        s[15] > 0 ? s[14] / s[15] * 0.1 : 0.0);
            }
        }
        dev_free(h->dd_prof);
        h->dd_prof = nullptr;
    }
#endif
    if (h->dd_prof) {                // experiment output: mean per-wave phase clocks of the LAST down-date
        const size_t cnt = (size_t)4096 * 4 * 4;
        std::vector<unsigned long long> v(cnt);
        if (hipMemcpy(v.data(), h->dd_prof, cnt * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            double s[4] = {0, 0, 0, 0}, mx = 0;
            size_t waves = 0;
            for (size_t w = 0; w < cnt / 4; ++w) {
                if (!v[4 * w + 3]) continue;
                ++waves;
                for (int i = 0; i < 4; ++i) s[i] += (double)v[4 * w + i];
                if ((double)v[4 * w + 3] > mx) mx = (double)v[4 * w + 3];
            }
            if (waves)
                fprintf(stderr, "[slamhip] down-date per-wave (mean of %zu waves): panel-wait %.0f clk  lifetime %.0f x 10 ns  epilogue %.0f clk  lifetime %.0f clk (max %.0f)\n",
                        waves, s[0] / waves, s[1] / waves, s[2] / waves, s[3] / waves, mx);
        }
        dev_free(h->dd_prof);
    }
    for (auto& p : h->pairs) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto& p : h->free_pairs) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    free_update_workspace(h);
    dev_free(h->x); dev_free(h->P); dev_free(h->Pside); dev_free(h->tiles);
    dev_free(h->obsbuf); dev_free(h->idfbuf); dev_free(h->d_assoc); dev_free(h->gate_part); dev_free(h->znbuf);
    dev_free(h->d_small); dev_free(h->d_status); dev_free(h->d_count); dev_free(h->d_pmax); dev_free(h->dd_claim);
    dev_free(h->grid_meta); dev_free(h->grid_cells); dev_free(h->grid_items);
    if (h->h_flag) (void)hipHostFree(h->h_flag);
    if (h->h_obs) (void)hipHostFree(h->h_obs);
    if (h->h_idf) (void)hipHostFree(h->h_idf);
    if (h->h_assoc) (void)hipHostFree(h->h_assoc);
    if (h->h_small) (void)hipHostFree(h->h_small);
    if (h->h_status) (void)hipHostFree(h->h_status);
    if (h->stage_ev) (void)hipEventDestroy(h->stage_ev);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return SLAM_OK;
}

static int create_impl(slam_ekf* h) {
    HIP_TRY(hipSetDevice(h->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, h->device));
    h->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIP_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    // ordering-only events between kernels of this device: no system-scope fence
    HIP_TRY(hipEventCreateWithFlags(&h->stage_ev, hipEventDisableTiming | hipEventDisableSystemFence));
    int rc;
    if ((rc = update_kernels_init())) return rc;
    if ((rc = gate_kernels_init())) return rc;
    if ((rc = dev_alloc_zero(&h->x, h->esz * (size_t)h->ncap, h->stream))) return rc;
    {   // tile-major, block lower (device_math.h): T (T + 1) / 2 tiles of E x E elements
        const int E = h->dtype == SLAM_F32 ? 128 : 64;
        const size_t T = (size_t)h->npad / E;
        if ((rc = dev_alloc_zero(&h->P, h->esz * (T * (T + 1) / 2) * E * E, h->stream))) return rc;
        if ((rc = dev_alloc_zero(&h->Pside, h->esz * 3 * (size_t)(h->npad / 2), h->stream))) return rc;
    }
    if ((rc = dev_alloc_zero(&h->d_small, sizeof(double) * 64, h->stream))) return rc;
    if ((rc = dev_alloc_zero(&h->d_count, sizeof(int32_t) * 4, h->stream))) return rc;
    if ((rc = dev_alloc_zero(&h->d_pmax, sizeof(double), h->stream))) return rc;
    if ((rc = dev_alloc_zero(&h->dd_claim, 8 * 64, h->stream))) return rc;
    HIP_TRY(hipHostMalloc((void**)&h->h_flag, sizeof(int32_t) * 16, hipHostMallocDefault));
    h->h_flag[0] = 0;
    HIP_TRY(hipHostGetDevicePointer((void**)&h->h_flag_dev, h->h_flag, 0));
    if ((h->debug_flags & 8) && (rc = dev_alloc_zero(&h->dd_prof, (size_t)4096 * 4 * 4 * 8, h->stream))) return rc;
#ifdef SLAMHIP_EXPERIMENTS
    if (!h->dd_prof && slam_exp_env("SLAMHIP_STAMPS", 0) && (rc = dev_alloc_zero(&h->dd_prof, (size_t)4096 * 16 * 8, h->stream))) return rc;
#endif
    if ((rc = dev_alloc_zero(&h->d_status, sizeof(int32_t) * 4, h->stream))) return rc;
    HIP_TRY(hipHostMalloc((void**)&h->h_small, sizeof(double) * 64, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void**)&h->h_status, sizeof(int32_t) * 4, hipHostMallocDefault));
    if ((rc = ensure_obs_capacity(h, 256))) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    return SLAM_OK;
}

extern "C" int slam_ekf_create(slam_ekf_t* out, int dtype, int max_landmarks, int device) {
    ARG_CHECK(out != nullptr, "handle pointer is null");
    *out = nullptr;
    ARG_CHECK(dtype == SLAM_F32 || dtype == SLAM_F64, "dtype must be SLAM_F32 or SLAM_F64");
    ARG_CHECK(max_landmarks >= 0 && max_landmarks <= 500000, "max_landmarks out of range");
    const int ndev = slam_device_count();
    if (ndev <= 0) {
        slam_set_error("no HIP device available: libslamhip has no CPU fallback");
        return SLAM_E_HIP;
    }
    ARG_CHECK(device >= 0 && device < ndev, "device index out of range");
    slam_ekf* h = new slam_ekf();
    h->dtype = dtype;
    h->device = device;
    h->maxN = max_landmarks;
    h->N = 0;
    h->ncap = 3 + 2 * max_landmarks;
    h->npad = round_up(h->ncap, SLAM_TILE);
    h->ld = h->npad;                  // P is allocated in whole tiles: npad x npad, padding stays zero
    h->esz = dtype == SLAM_F32 ? 4 : 8;
    h->x = h->P = h->Pside = nullptr;
    h->stream = nullptr;
    h->stage_ev = nullptr; h->stage_pending = 0;
    h->PHtS = nullptr;
    h->kcap = 0;
    h->W1 = h->W2 = h->Wimg = nullptr;
    h->PHt = h->PHtS = h->Kd = h->Cmat = h->Smat = h->Mwork = h->gvec = nullptr;
    h->tiles = nullptr; h->tiles_T = h->tiles_len = h->tiles_cap = 0; h->tilesB_off = h->tilesB_len = 0;
    h->obsbuf = nullptr; h->idfbuf = nullptr; h->ocap = 0;
    h->h_obs = nullptr; h->h_idf = nullptr; h->h_assoc = nullptr; h->d_assoc = nullptr;
    h->gate_part = nullptr; h->gate_blocks_cap = 0;
    h->d_small = h->h_small = nullptr;
    h->d_pmax = nullptr; h->pmax_valid = 0; h->dd_claim = nullptr;
    h->gate_mode = SLAM_GATE_AUTO; h->gate_last = 0; h->grid_meta = nullptr; h->grid_cells = nullptr; h->grid_items = nullptr;
    h->grid_force = 1; h->grid_upd = 0; h->grid_live = 0; h->grid_n_seen = 0;
    h->znbuf = nullptr; h->d_count = nullptr; h->h_flag = nullptr; h->h_flag_dev = nullptr; h->obs_seq = 0;
    h->d_status = h->h_status = nullptr;
    h->async_updates = 0; h->deferred = 0; h->pending_status = 0; h->debug_stamps = 0;
    h->debug_flags = 0;
#ifdef SLAMHIP_EXPERIMENTS
    // timing experiments on the down-date that give WRONG RESULTS: only in the separate `make exp` library
    h->debug_flags = getenv("SLAMHIP_DEBUG") ? atoi(getenv("SLAMHIP_DEBUG")) : 0;
#endif
    h->dd_prof = nullptr;
    h->xflags = (getenv("SLAMHIP_X") ? atoi(getenv("SLAMHIP_X")) : 0) & SLAM_XFLAGS_MASK;
    h->factor_blocked = 1;
#ifdef SLAMHIP_EXPERIMENTS
    h->factor_blocked = !(getenv("SLAMHIP_FACTOR") && !strcmp(getenv("SLAMHIP_FACTOR"), "scalar"));
    if (getenv("SLAMHIP_FACTOR") && !strcmp(getenv("SLAMHIP_FACTOR"), "pivot1")) h->factor_blocked = 2;     // round 3's one pivot per MFMA
#endif
    h->timing = 0;
    for (int i = 0; i < SLAM_K_COUNT; ++i) { h->t_ms[i] = 0; h->t_sq[i] = 0; h->t_n[i] = 0; h->t_min[i] = 0; }
    const int rc = create_impl(h);
    if (rc != SLAM_OK) {
        slam_ekf_destroy(h);
        return rc;
    }
    *out = h;
    return SLAM_OK;
}

// ---- state I/O -------------------------------------------------------------------
// columns per band of the staged state transfers: a multiple of the tile edge, n x W elements <= 256 MiB
static int state_band_columns(const slam_ekf* h, int n) {
    const size_t budget = (size_t)256 << 20;
    size_t w = budget / (h->esz * (size_t)n) / SLAM_TILE * SLAM_TILE;
    if (w < SLAM_TILE) w = SLAM_TILE;
    if (w > (size_t)h->npad) w = (size_t)h->npad;
    return (int)w;
}

static int set_state_impl(slam_ekf* h, const void* x, const void* P, int n, int ldP, hipMemcpyKind kind) {
    ARG_CHECK(h != nullptr, "null handle");
    ARG_CHECK(x != nullptr && P != nullptr, "x / P is null");
    ARG_CHECK(n >= 3 && ((n - 3) % 2) == 0, "n must be 3 + 2*N");
    ARG_CHECK(ldP >= n, "ldP < n");
    const int N = (n - 3) / 2;
    if (N > h->maxN) {
        slam_set_error("state has %d landmarks, capacity is %d", N, h->maxN);
        return SLAM_E_CAPACITY;
    }
    HIP_TRY(hipSetDevice(h->device));
    if (N < h->N) {                 // panel rows >= n must be zero for the down-date
        const int rc = zero_panels(h);
        if (rc) return rc;
    }
    HIP_TRY(hipMemcpyAsync(h->x, x, h->esz * (size_t)n, kind, h->stream));
    // the caller's matrix is column-major (Julia order); the state is tile-major: a device-side repack.  A host source is
    // staged band by band through a BOUNDED column-major device buffer (<= 256 MiB: the 80 GB matrix of N = 50k fp64 must
    // not need another 80 GB of scratch), a device source is packed where it lies.
    int rc = SLAM_OK;
    hipError_t es = hipSuccess;
    if (kind == hipMemcpyHostToDevice) {
        const int W = state_band_columns(h, n);
        void* d_tmp = nullptr;
        HIP_TRY(hipMalloc(&d_tmp, h->esz * (size_t)n * W));
        for (int c = 0; c < h->npad && rc == SLAM_OK && es == hipSuccess; c += W) {
            const int w = c + W <= h->npad ? W : h->npad - c;
            const int wn = n - c < 0 ? 0 : (n - c < w ? n - c : w);           // real columns of this band (the rest is padding)
            if (wn > 0)
                es = hipMemcpy2DAsync(d_tmp, h->esz * (size_t)n, (const char*)P + h->esz * (size_t)c * ldP, h->esz * (size_t)ldP,
                                      h->esz * (size_t)n, (size_t)wn, kind, h->stream);
            if (es == hipSuccess) rc = launch_pack(h, d_tmp, n, n, c, w);
            if (es == hipSuccess && rc == SLAM_OK) es = hipStreamSynchronize(h->stream);   // the staging buffer is reused
        }
        (void)hipFree(d_tmp);
    } else {
        rc = launch_pack(h, P, ldP, n, 0, h->npad);
        es = hipStreamSynchronize(h->stream);
    }
    if (rc) return rc;
    if (es != hipSuccess) {
        slam_set_error("upload of the state failed: %s", hipGetErrorString(es));
        return SLAM_E_HIP;
    }
    h->N = N;
    h->pmax_valid = 0;                  // the pre-gate's variance bound belongs to the old matrix
    h->grid_force = 1;                  // ... and the grid to the old means
    return SLAM_OK;
}

extern "C" int slam_ekf_set_state(slam_ekf_t h, const void* x, const void* P, int n, int ldP) {
    SLAM_RANGE();
    return set_state_impl(h, x, P, n, ldP, hipMemcpyHostToDevice);
}

extern "C" int slam_ekf_set_state_device(slam_ekf_t h, const void* d_x, const void* d_P, int n, int ldP) {
    SLAM_RANGE();
    return set_state_impl(h, d_x, d_P, n, ldP, hipMemcpyDeviceToDevice);
}

extern "C" int slam_ekf_get_state(slam_ekf_t h, void* x, void* P, int n, int ldP) {
    SLAM_RANGE();
    ARG_CHECK(h != nullptr, "null handle");
    ARG_CHECK(n == 3 + 2 * h->N, "n does not match the state (3 + 2*N)");
    ARG_CHECK(P == nullptr || ldP >= n, "ldP < n");
    HIP_TRY(hipSetDevice(h->device));
    if (x) HIP_TRY(hipMemcpyAsync(x, h->x, h->esz * (size_t)n, hipMemcpyDeviceToHost, h->stream));
    int rc = SLAM_OK;
    hipError_t es = hipSuccess;
    if (P) {
        // the full symmetric matrix in the caller's column-major order: unpacked on the device from the stored triangle,
        // band by band through a bounded staging buffer (<= 256 MiB)
        const int W = state_band_columns(h, n);
        void* d_tmp = nullptr;
        HIP_TRY(hipMalloc(&d_tmp, h->esz * (size_t)n * W));
        for (int c = 0; c < n && rc == SLAM_OK && es == hipSuccess; c += W) {
            const int w = c + W <= n ? W : n - c;
            rc = launch_unpack(h, d_tmp, n, n, c, w);
            if (rc == SLAM_OK)
                es = hipMemcpy2DAsync((char*)P + h->esz * (size_t)c * ldP, h->esz * (size_t)ldP, d_tmp, h->esz * (size_t)n,
                                      h->esz * (size_t)n, (size_t)w, hipMemcpyDeviceToHost, h->stream);
            if (rc == SLAM_OK && es == hipSuccess) es = hipStreamSynchronize(h->stream);   // the staging buffer is reused
        }
        (void)hipFree(d_tmp);
    }
    if (es == hipSuccess) es = hipStreamSynchronize(h->stream);
    if (rc) return rc;
    if (es != hipSuccess) {
        slam_set_error("download of the state failed: %s", hipGetErrorString(es));
        return SLAM_E_HIP;
    }
    return SLAM_OK;
}

// A block / the diagonal of the covariance without downloading it (80 GB at N = 50k): gathered on the device through
// the symmetric view, one dense copy to the host.
static int get_gathered(slam_ekf* h, int r0, int c0, int nr, int nc, int diag, void* out, int ld_out) {
    HIP_TRY(hipSetDevice(h->device));
    void* d_tmp = nullptr;
    HIP_TRY(hipMalloc(&d_tmp, h->esz * (size_t)nr * nc));
    int rc = launch_block_gather(h, r0, c0, nr, nc, diag, d_tmp);
    if (rc == SLAM_OK) {
        hipError_t e = hipMemcpy2DAsync(out, h->esz * (size_t)ld_out, d_tmp, h->esz * (size_t)nr, h->esz * (size_t)nr, (size_t)nc,
                                        hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) {
            slam_set_error("HIP error while reading a covariance block: %s", hipGetErrorString(e));
            rc = SLAM_E_HIP;
        }
    }
    (void)hipFree(d_tmp);
    return rc;
}

extern "C" int slam_ekf_get_block(slam_ekf_t h, int r0, int c0, int nr, int nc, void* out, int ld_out) {
    SLAM_RANGE();
    ARG_CHECK(h != nullptr, "null handle");
    const int n = 3 + 2 * h->N;
    ARG_CHECK(nr >= 0 && nc >= 0, "negative block size");
    if (nr == 0 || nc == 0) return SLAM_OK;
    ARG_CHECK(out != nullptr, "out is null");
    ARG_CHECK(r0 >= 0 && c0 >= 0 && r0 + nr <= n && c0 + nc <= n, "block outside the n x n covariance (Julia: BoundsError)");
    ARG_CHECK(ld_out >= nr, "ld_out < nr");
    ARG_CHECK((size_t)nr * nc <= ((size_t)1 << 28), "block larger than 2^28 elements: read it in pieces");
    return get_gathered(h, r0, c0, nr, nc, 0, out, ld_out);
}

extern "C" int slam_ekf_get_diag(slam_ekf_t h, void* out) {
    ARG_CHECK(h != nullptr && out != nullptr, "null argument");
    const int n = 3 + 2 * h->N;
    return get_gathered(h, 0, 0, n, 1, 1, out, n);
}

/* The landmarks' 2 x 2 covariance blocks, packed: out[0][j] = P[f, f], out[1][j] = P[f+1, f], out[2][j] = P[f+1, f+1]
 * (f = 3 + 2 j, j = 0 .. N-1; three rows of N values in the handle's dtype) -- straight from the side array the gating
 * sweep reads (device_math.h: side_note), which every writer of those entries keeps. */
extern "C" int slam_ekf_get_landmark_blocks(slam_ekf_t h, void* out) {
    ARG_CHECK(h != nullptr && out != nullptr, "null argument");
    if (h->N == 0) return SLAM_OK;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpy2DAsync(out, h->esz * (size_t)h->N, h->Pside, h->esz * (size_t)(h->npad / 2), h->esz * (size_t)h->N, 3,
                             hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return SLAM_OK;
}

extern "C" int slam_ekf_get_pose(slam_ekf_t h, double pose[3]) {
    ARG_CHECK(h != nullptr && pose != nullptr, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(h->h_small, h->x, h->esz * 3, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int i = 0; i < 3; ++i)
        pose[i] = h->dtype == SLAM_F32 ? (double)((float*)h->h_small)[i] : h->h_small[i];
    return SLAM_OK;
}

/* feature_ellipses(x, cov) and the vehicle ellipse of monitor()  (sim/browser/wsserver.jl:60-65,72-85). */
extern "C" int slam_ekf_ellipses(slam_ekf_t h, double* features, double vehicle[6]) {
    SLAM_RANGE();
    ARG_CHECK(h != nullptr, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    const size_t cnt = (size_t)h->N + 1;
    double* d_out = nullptr;
    HIP_TRY(hipMalloc((void**)&d_out, sizeof(double) * 5 * cnt));
    std::vector<double> host(5 * cnt);
    int rc = launch_ellipses(h, d_out);
    if (rc == SLAM_OK) {
        const hipError_t e1 = hipMemcpyAsync(host.data(), d_out, sizeof(double) * 5 * cnt, hipMemcpyDeviceToHost, h->stream);
        const hipError_t e2 = e1 == hipSuccess ? hipStreamSynchronize(h->stream) : e1;
        if (e2 != hipSuccess) {
            slam_set_error("HIP error in slam_ekf_ellipses: %s", hipGetErrorString(e2));
            rc = SLAM_E_HIP;
        }
    }
    (void)hipFree(d_out);
    if (rc) return rc;
    if (features)
        for (size_t i = 0; i < 5 * (cnt - 1); ++i) features[i] = host[5 + i];
    if (vehicle) {
        double pose[3];
        if ((rc = slam_ekf_get_pose(h, pose))) return rc;
        vehicle[0] = pose[0]; vehicle[1] = pose[1]; vehicle[2] = pose[2];      // cx, cy, vehicle_phi
        vehicle[3] = host[2]; vehicle[4] = host[3]; vehicle[5] = host[4];      // rx, ry, phi
    }
    return SLAM_OK;
}

extern "C" int slam_ekf_num_landmarks(slam_ekf_t h, int* N) {
    ARG_CHECK(h != nullptr && N != nullptr, "null argument");
    *N = h->N;
    return SLAM_OK;
}

extern "C" int slam_ekf_dtype(slam_ekf_t h, int* dtype) {
    ARG_CHECK(h != nullptr && dtype != nullptr, "null argument");
    *dtype = h->dtype;
    return SLAM_OK;
}

extern "C" int slam_ekf_device_ptrs(slam_ekf_t h, void** d_x, void** d_P, int* ld, void** stream) {
    ARG_CHECK(h != nullptr, "null handle");
    if (d_x) *d_x = h->x;
    if (d_P) *d_P = h->P;
    if (ld) *ld = h->ld;
    if (stream) *stream = (void*)h->stream;
    return SLAM_OK;
}

extern "C" int slam_ekf_state_written(slam_ekf_t h) {
    SLAM_RANGE();
    ARG_CHECK(h != nullptr, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    h->pmax_valid = 0;                  // the pre-gate's variance bound is recomputed before the next sweep
    h->grid_force = 1;                  // the grid of landmark means is rebuilt at the next query
    return launch_side_rebuild(h);      // the packed 2 x 2 diagonal blocks follow the matrix again
}

extern "C" int slam_ekf_copy_floor(slam_ekf_t h, int reps, double out[2]) {
    SLAM_RANGE();
    ARG_CHECK(h != nullptr && out != nullptr, "null argument");
    ARG_CHECK(reps >= 1 && reps <= 1000, "reps out of range");
    HIP_TRY(hipSetDevice(h->device));
    return launch_copy_floor(h, reps, out);
}

// ---- hot path ----------------------------------------------------------------------
static int check_R(const double* R) {
    ARG_CHECK(R != nullptr, "R is null");
    return SLAM_OK;
}

// stage nobs observation pairs (+ optional idf) on the device, stream-ordered
static int stage_obs(slam_ekf* h, const double* z, const int32_t* idf, int nobs) {
    int rc = ensure_obs_capacity(h, nobs);
    if (rc) return rc;
    // the pinned staging buffers may still be the source of an earlier async copy
    if (h->stage_pending) {
        HIP_TRY(hipEventSynchronize(h->stage_ev));
        h->stage_pending = 0;
    }
    memcpy(h->h_obs, z, sizeof(double) * 2 * (size_t)nobs);
    HIP_TRY(hipMemcpyAsync(h->obsbuf, h->h_obs, sizeof(double) * 2 * (size_t)nobs, hipMemcpyHostToDevice, h->stream));
    if (idf) {
        memcpy(h->h_idf, idf, sizeof(int32_t) * (size_t)nobs);
        HIP_TRY(hipMemcpyAsync(h->idfbuf, h->h_idf, sizeof(int32_t) * (size_t)nobs, hipMemcpyHostToDevice, h->stream));
    }
    HIP_TRY(hipEventRecord(h->stage_ev, h->stream));
    h->stage_pending = 1;
    return SLAM_OK;
}

extern "C" int slam_ekf_predict(slam_ekf_t h, double v, double g, double wheelbase, const double Q[4], double dt) {
    SLAM_RANGE();
    ARG_CHECK(h != nullptr && Q != nullptr, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    return launch_predict(h, v, g, wheelbase, Q, dt);
}

extern "C" int slam_ekf_associate(slam_ekf_t h, const double* z, int nz, const double R[4], double gate1, double gate2,
                                  int32_t* assoc) {
    SLAM_RANGE();
    ARG_CHECK(h != nullptr, "null handle");
    ARG_CHECK(nz >= 0, "nz < 0");
    if (nz == 0) return SLAM_OK;
    ARG_CHECK(z != nullptr && assoc != nullptr, "z / assoc is null");
    int rc = check_R(R);
    if (rc) return rc;
    if (h->N == 0) {                       // outer stays Inf > gate2: every observation is a new feature
        for (int i = 0; i < nz; ++i) assoc[i] = -1;
        return SLAM_OK;
    }
    HIP_TRY(hipSetDevice(h->device));
    if ((rc = ensure_obs_capacity(h, nz))) return rc;
    if ((rc = launch_gate(h, nz, R, gate1, gate2, z, false))) return rc;       // (z travels in the kernel arguments)
    HIP_TRY(hipMemcpyAsync(h->h_assoc, h->d_assoc, sizeof(int32_t) * (size_t)nz, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    memcpy(assoc, h->h_assoc, sizeof(int32_t) * (size_t)nz);
    return SLAM_OK;
}

extern "C" int slam_ekf_set_gate_mode(slam_ekf_t h, int mode) {
    ARG_CHECK(h != nullptr, "null handle");
    ARG_CHECK(mode == SLAM_GATE_AUTO || mode == SLAM_GATE_SWEEP || mode == SLAM_GATE_GRID, "unknown gate mode");
    h->gate_mode = mode;
    h->grid_force = 1;          // (also the way to say "the means were written through the raw device view")
    return SLAM_OK;
}

extern "C" int slam_ekf_gate_info(slam_ekf_t h, int64_t out[8]) {
    ARG_CHECK(h != nullptr && out != nullptr, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    return gate_info(h, out);
}

extern "C" int slam_ekf_nis(slam_ekf_t h, const double z1[2], int j, const double R[4], double out[2]) {
    ARG_CHECK(h != nullptr && z1 != nullptr && out != nullptr, "null argument");
    int rc = check_R(R);
    if (rc) return rc;
    ARG_CHECK(j >= 1 && j <= h->N, "landmark index out of range");
    HIP_TRY(hipSetDevice(h->device));
    if ((rc = launch_nis(h, z1, j, R))) return rc;
    HIP_TRY(hipMemcpyAsync(h->h_small, h->d_small, sizeof(double) * 2, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    out[0] = h->h_small[0];
    out[1] = h->h_small[1];
    return SLAM_OK;
}

extern "C" int slam_ekf_predict_observation(slam_ekf_t h, int j, double zp[2], double Hv[6], double Hf[4]) {
    ARG_CHECK(h != nullptr && zp != nullptr && Hv != nullptr && Hf != nullptr, "null argument");
    ARG_CHECK(j >= 1 && j <= h->N, "landmark index out of range");
    HIP_TRY(hipSetDevice(h->device));
    int rc;
    if ((rc = launch_obs_model(h, j))) return rc;
    HIP_TRY(hipMemcpyAsync(h->h_small, h->d_small, sizeof(double) * 12, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    zp[0] = h->h_small[0]; zp[1] = h->h_small[1];
    for (int i = 0; i < 6; ++i) Hv[i] = h->h_small[2 + i];
    for (int i = 0; i < 4; ++i) Hf[i] = h->h_small[8 + i];
    return SLAM_OK;
}

static int read_status(slam_ekf* h, int sticky) {
    HIP_TRY(hipMemcpyAsync(h->h_status, h->d_status, sizeof(int32_t) * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const int bad = sticky ? h->h_status[1] : h->h_status[0];
    if (h->h_status[1]) HIP_TRY(hipMemsetAsync(h->d_status + 1, 0, sizeof(int32_t), h->stream));
    h->pending_status = 0;
    if (bad == 2) {      // (factor_w1_kernel: a panel wave gave up waiting for the factorising workgroup -- a defect, not a property of the data)
        slam_set_error("update: the factorisation did not publish within 2 s; covariance unchanged, the mean may be partly updated");
        return SLAM_E_HIP;
    }
    if (bad) {
        slam_set_error("innovation covariance S is not positive definite (state left unchanged)");
        return SLAM_E_NOTPD;
    }
    return SLAM_OK;
}

extern "C" int slam_ekf_update(slam_ekf_t h, const double* zf, const int32_t* idf, int m, const double R[4], int form) {
    SLAM_RANGE();
    ARG_CHECK(h != nullptr, "null handle");
    ARG_CHECK(m >= 0, "m < 0");
    ARG_CHECK(form == SLAM_FORM_CHOLESKY || form == SLAM_FORM_JOSEPH, "unknown update form");
    if (m == 0) return SLAM_OK;            // H is 0 x n: the reference's update is the identity
    ARG_CHECK(zf != nullptr && idf != nullptr, "zf / idf is null");
    int rc = check_R(R);
    if (rc) return rc;
    for (int i = 0; i < m; ++i) ARG_CHECK(idf[i] >= 1 && idf[i] <= h->N, "idf entry out of range (Julia: BoundsError)");
    HIP_TRY(hipSetDevice(h->device));
    if ((rc = ensure_update_workspace(h, m))) return rc;
    if ((rc = stage_obs(h, zf, idf, m))) return rc;
    if ((rc = launch_update(h, m, R, form, false))) return rc;
    if (h->async_updates) {
        h->pending_status = 1;
        return SLAM_OK;
    }
    return read_status(h, 0);
}

extern "C" int slam_ekf_augment(slam_ekf_t h, const double* zn, int nn, const double R[4]) {
    SLAM_RANGE();
    ARG_CHECK(h != nullptr, "null handle");
    ARG_CHECK(nn >= 0, "nn < 0");
    if (nn == 0) return SLAM_OK;
    ARG_CHECK(zn != nullptr, "zn is null");
    int rc = check_R(R);
    if (rc) return rc;
    if (h->N + nn > h->maxN) {
        slam_set_error("augment: %d + %d landmarks exceed capacity %d", h->N, nn, h->maxN);
        return SLAM_E_CAPACITY;
    }
    HIP_TRY(hipSetDevice(h->device));
    if ((rc = stage_obs(h, zn, nullptr, nn))) return rc;
    if ((rc = launch_augment(h, nn, R, h->obsbuf))) return rc;
    h->N += nn;
    return SLAM_OK;
}

/* One observation step of the reference's sim! loop without a host round trip between its parts:
 * associate (data-association.jl:1-51) -> update (ekf.jl:46-77) -> add_features (ekf.jl:84-122).
 * The association vector is compacted ON THE DEVICE into the update's inputs; the update kernels are launched with
 * nz as an upper bound and read the matched count from d_count, so they are queued while the gating still runs.
 * The host waits only for the association vector (an event recorded BEFORE the update kernels) to learn how many
 * new features to append. */
extern "C" int slam_ekf_observe(slam_ekf_t h, const double* z, int nz, const double R[4], double gate1, double gate2, int form,
                                int32_t* assoc) {
    SLAM_RANGE();
    ARG_CHECK(h != nullptr, "null handle");
    ARG_CHECK(nz >= 0, "nz < 0");
    ARG_CHECK(form == SLAM_FORM_CHOLESKY || form == SLAM_FORM_JOSEPH, "unknown update form");
    if (nz == 0) return SLAM_OK;
    ARG_CHECK(z != nullptr && assoc != nullptr, "z / assoc is null");
    int rc = check_R(R);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    if (h->N == 0) {                       // every observation is a new feature (outer stays Inf > gate2)
        for (int i = 0; i < nz; ++i) assoc[i] = -1;
        return slam_ekf_augment(h, z, nz, R);
    }
    if ((rc = ensure_update_workspace(h, nz))) return rc;
    if ((rc = ensure_obs_capacity(h, nz))) return rc;
    // No staging copy: z travels in the gating kernel's arguments (its first workgroup leaves the device copy the
    // compaction, the update and add_features read); the decisions come back through pinned host memory.
    h->obs_seq = h->obs_seq == 0x7fffffff ? 1 : h->obs_seq + 1;
    if ((rc = launch_gate(h, nz, R, gate1, gate2, z, true))) return rc;      // gating + compaction
    if ((rc = launch_update(h, nz, R, form, true))) return rc;
    // Wait for the decisions only (the update runs on behind them): the compaction publishes this call's sequence
    // number in pinned memory after the association vector.  The stream is queried now and then so that a failed
    // kernel cannot leave the host spinning.
    {
        volatile int32_t* flag = h->h_flag;
        unsigned long long spins = 0;
        while (*flag != h->obs_seq) {
            __builtin_ia32_pause();
            if ((++spins & 0xfffffull) == 0) {
                const hipError_t q = hipStreamQuery(h->stream);
                if (q != hipErrorNotReady && *flag != h->obs_seq) {
                    if (q == hipSuccess) slam_set_error("observe: the gating finished without publishing its decisions");
                    else slam_set_error("HIP error while waiting for the gating: %s", hipGetErrorString(q));
                    return SLAM_E_HIP;
                }
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    int nn = 0;
    for (int i = 0; i < nz; ++i) {
        assoc[i] = h->h_assoc[i];
        nn += assoc[i] < 0;
    }
    int cap_rc = SLAM_OK;
    if (nn) {
        if (h->N + nn > h->maxN) cap_rc = SLAM_E_CAPACITY;          // reported below, behind the update's own status
        else {
            if ((rc = launch_augment(h, nn, R, h->znbuf))) return rc;
            h->N += nn;
        }
    }
    // The update is already queued: its status is collected (or left pending in async mode) whatever happened to
    // the new features.  S not positive definite outranks the capacity overflow: the caller must learn that the
    // update was NOT applied.
    if (h->async_updates) h->pending_status = 1;
    else if ((rc = read_status(h, 0))) return rc;
    if (cap_rc) {
        slam_set_error("observe: %d + %d landmarks exceed capacity %d (the update was applied, no feature added)", h->N, nn,
                       h->maxN);
        return cap_rc;
    }
    return SLAM_OK;
}

// ---- stream / timing -----------------------------------------------------------------
#ifdef SLAMHIP_EXPERIMENTS
/* Experiments build only (tools/graph_observe.py): the observation step ENQUEUED and nothing else -- gating + compaction,
 * then the update with the matched count read on the device -- without waiting for the decisions and without
 * add_features, i.e. valid only for a workload in which no new feature arises.  It exists so that the six launches of a
 * step can be captured into a hipGraph (a capture cannot contain the host's wait for the decisions). */
extern "C" int slam_exp_observe_enqueue(slam_ekf_t h, const double* z, int nz, const double R[4], double gate1, double gate2) {
    ARG_CHECK(h != nullptr && z != nullptr && nz > 0 && h->N > 0, "bad argument");
    int rc = check_R(R);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(h->device));
    if ((rc = ensure_update_workspace(h, nz))) return rc;
    if ((rc = ensure_obs_capacity(h, nz))) return rc;
    h->obs_seq = h->obs_seq == 0x7fffffff ? 1 : h->obs_seq + 1;
    if ((rc = launch_gate(h, nz, R, gate1, gate2, z, true))) return rc;
    return launch_update(h, nz, R, SLAM_FORM_CHOLESKY, true);
}

/* Experiments build only (tools/dbg_fw1.py): the update's workspace as it stands -- what: 0 = W1 ([npad][2 kcap], handle dtype),
 * 1 = the pre-split bf16 image of W1 (fp32 handles), 2 = C ([kcap][kcap] doubles), 3 = g ([kcap] doubles).  Copies
 * min(bytes, size) bytes; *size receives the buffer's size. */
extern "C" int slam_exp_workspace(slam_ekf_t h, int what, void* out, size_t bytes, size_t* size) {
    ARG_CHECK(h != nullptr && h->kcap > 0, "no workspace yet");
    HIP_TRY(hipSetDevice(h->device));
    const void* src = nullptr;
    size_t sz = 0;
    if (what == 0) { src = h->W1; sz = h->esz * (size_t)h->npad * 2 * h->kcap; }
    else if (what == 1) { src = h->Wimg; sz = h->Wimg ? (size_t)(h->npad / 128) * (h->kcap / 16) * 3 * 4096 : 0; }
    else if (what == 2) { src = h->Cmat; sz = sizeof(double) * (size_t)h->kcap * h->kcap; }
    else if (what == 3) { src = h->gvec; sz = sizeof(double) * (size_t)h->kcap; }
    ARG_CHECK(src != nullptr, "no such buffer");
    if (size) *size = sz;
    if (out && bytes) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        HIP_TRY(hipMemcpy(out, src, bytes < sz ? bytes : sz, hipMemcpyDeviceToHost));
    }
    return SLAM_OK;
}
#endif

extern "C" int slam_ekf_set_async(slam_ekf_t h, int async_updates) {
    ARG_CHECK(h != nullptr, "null handle");
    h->async_updates = async_updates ? 1 : 0;
    return SLAM_OK;
}

extern "C" int slam_ekf_sync(slam_ekf_t h) {
    SLAM_RANGE();
    ARG_CHECK(h != nullptr, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    if (h->pending_status) return read_status(h, 1);
    HIP_TRY(hipStreamSynchronize(h->stream));
    return SLAM_OK;
}

extern "C" int slam_ekf_debug_stamps(slam_ekf_t h, int enable, uint64_t* out16) {
    ARG_CHECK(h != nullptr, "null handle");
    h->debug_stamps = enable ? 1 : 0;
    if (out16) {      // [0..7]: the factorisation's workgroup; [8..15]: the first panel workgroup of factor_w1_kernel
        HIP_TRY(hipSetDevice(h->device));
        HIP_TRY(hipMemcpyAsync(h->h_small, h->d_small + 40, sizeof(uint64_t) * 16, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        memcpy(out16, h->h_small, sizeof(uint64_t) * 16);
    }
    return SLAM_OK;
}

extern "C" int slam_ekf_timing(slam_ekf_t h, int enable) {
    ARG_CHECK(h != nullptr, "null handle");
    h->timing = enable < 0 ? 0 : enable;
    return SLAM_OK;
}

extern "C" int slam_ekf_timing_read(slam_ekf_t h, int kid, double* total_ms, int64_t* launches) {
    ARG_CHECK(h != nullptr, "null handle");
    ARG_CHECK(kid >= 0 && kid < SLAM_K_COUNT, "kernel id out of range");
    HIP_TRY(hipSetDevice(h->device));
    const int rc = fold_timing(h);
    if (rc) return rc;
    if (total_ms) *total_ms = h->t_ms[kid];
    if (launches) *launches = h->t_n[kid];
    return SLAM_OK;
}

extern "C" int slam_ekf_timing_min(slam_ekf_t h, int kid, double* min_ms) {
    ARG_CHECK(h != nullptr && min_ms != nullptr, "null argument");
    ARG_CHECK(kid >= 0 && kid < SLAM_K_COUNT, "kernel id out of range");
    HIP_TRY(hipSetDevice(h->device));
    const int rc = fold_timing(h);
    if (rc) return rc;
    *min_ms = h->t_min[kid];
    return SLAM_OK;
}

extern "C" int slam_ekf_timing_stats(slam_ekf_t h, int kid, double out[4]) {
    ARG_CHECK(h != nullptr && out != nullptr, "null argument");
    ARG_CHECK(kid >= 0 && kid < SLAM_K_COUNT, "kernel id out of range");
    HIP_TRY(hipSetDevice(h->device));
    const int rc = fold_timing(h);
    if (rc) return rc;
    const double nn = (double)h->t_n[kid];
    const double mean = nn > 0 ? h->t_ms[kid] / nn : 0.0;
    const double var = nn > 1 ? (h->t_sq[kid] - nn * mean * mean) / (nn - 1.0) : 0.0;
    out[0] = nn; out[1] = mean; out[2] = var > 0 ? sqrt(var) : 0.0; out[3] = h->t_min[kid];
    return SLAM_OK;
}

extern "C" int slam_ekf_timing_reset(slam_ekf_t h) {
    ARG_CHECK(h != nullptr, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    const int rc = fold_timing(h);
    if (rc) return rc;
    for (int i = 0; i < SLAM_K_COUNT; ++i) { h->t_ms[i] = 0; h->t_sq[i] = 0; h->t_n[i] = 0; h->t_min[i] = 0; }
    return SLAM_OK;
}
