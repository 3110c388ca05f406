// ekf_update.hip -- K2..K5: the front half of update()  (src/ekf.jl:46-77).
//
//   K2/K3  pht_kernel      PHt = P*H'  from the 5 non-zero columns of each H row pair
//                          (the reference multiplies by a dense 2m x n H, :67)
//   K4     factor_kernel   S = H*PHt + RR, S = (S+S')/2, C = inv(chol(S))   (:68-70)
//                          one workgroup, double precision, LDS resident
//   K5     panel_gemm      W1 = PHt*C (:71);   x += W*v = PHt*(C*C'*v)   (:72,:74)
//
// The covariance down-date P -= W1*W1' (:75) is in ekf_syrk.hip.
//
// Panel buffers are ROW-major [npad][pitch] in the state dtype, zero in rows >= n
// (never written) and in columns k..kp-1 (written as zeros here), so the
// down-date needs no guards on its operand loads.
#include "common.h"
#include "device_math.h"

int launch_downdate(slam_ekf* h, int kp_total, const void* X, const void* Y, int pitch);   // ekf_syrk.hip
int joseph_T_pass(slam_ekf* h, int kp);

namespace {

// ---------------------------------------------------------------------------
// K2/K3: PHt[r, 2i:2i+2] = P[r,0:3]*Hv_i' + P[r,f_i:f_i+2]*Hf_i'
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pht_kernel(const T* __restrict__ x, const T* __restrict__ P, int ld, int n,
                                                   const int32_t* __restrict__ idf, int m, int k, int kp,
                                                   T* __restrict__ PHt, int pitch) {
    extern __shared__ double sm[];            // [m][10] Jacobian blocks, then int f[m]
    int* sf = reinterpret_cast<int*>(sm + (size_t)10 * m);
    const int tid = threadIdx.x;
    const double xv = (double)x[0], yv = (double)x[1], phi = (double)x[2];
    for (int i = tid; i < m; i += blockDim.x) {
        const int f = 3 + 2 * (idf[i] - 1);
        const ObsModel om = obs_model(xv, yv, phi, (double)x[f], (double)x[f + 1]);
#pragma unroll
        for (int q = 0; q < 6; ++q) sm[10 * i + q] = om.Hv[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) sm[10 * i + 6 + q] = om.Hf[q];
        sf[i] = f;
    }
    __syncthreads();
    const int r = blockIdx.x * blockDim.x + tid;
    if (r >= n) return;
    const double p0 = (double)P[(size_t)0 * ld + r];
    const double p1 = (double)P[(size_t)1 * ld + r];
    const double p2 = (double)P[(size_t)2 * ld + r];
    T* out = PHt + (size_t)r * pitch;
    for (int i = 0; i < m; ++i) {
        const double* hb = sm + 10 * i;
        const int f = sf[i];
        const double q0 = (double)P[(size_t)f * ld + r];
        const double q1 = (double)P[(size_t)(f + 1) * ld + r];
        out[2 * i] = (T)(hb[0] * p0 + hb[1] * p1 + hb[2] * p2 + hb[6] * q0 + hb[7] * q1);
        out[2 * i + 1] = (T)(hb[3] * p0 + hb[4] * p1 + hb[5] * p2 + hb[8] * q0 + hb[9] * q1);
    }
    for (int c = k; c < kp; ++c) out[c] = (T)0;
}

// ---------------------------------------------------------------------------
// K4: one workgroup builds S, factors it and emits C = inv(chol(S)).
//
// Factorisation: in-place Gauss-Jordan elimination without pivoting on the SPD
// matrix (an LDL' factorisation): after step j the strictly-lower part of row i
// holds row i of inv(L) (unit lower), the diagonal holds D.  Then
//   chol(S) = U = sqrt(D) L'   and   C = inv(U) = inv(L)' / sqrt(D)  (upper),
// which is the reference's inv(chol(S)) (unique: upper, positive diagonal,
// C*C' = inv(S)).  Two barriers per elimination step.
// ---------------------------------------------------------------------------
constexpr int FACTOR_THREADS = 1024;

template <typename T>
__global__ __launch_bounds__(FACTOR_THREADS) void factor_kernel(
    const T* __restrict__ x, const T* __restrict__ PHt, int pht_pitch, const double* __restrict__ z,
    const int32_t* __restrict__ idf, int m, int k, int kp, double R0, double R1, double R2, double R3,
    T* __restrict__ Cout, int c_pitch, double* __restrict__ gvec, double* __restrict__ Sout, int want_sinv,
    double* __restrict__ Mglobal, int32_t* __restrict__ status) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x;
    const int nt = blockDim.x;
    const int mp = kp + 1;                                   // pitch of M (odd: conflict-free column walks)
    double* M = Mglobal ? Mglobal : lds;
    double* aux = Mglobal ? lds : lds + (size_t)kp * mp;
    double* mvec = aux;                                      // [kp]
    double* vvec = aux + kp;                                 // [kp] innovation
    double* yvec = aux + 2 * kp;                             // [kp]
    double* hb = aux + 3 * kp;                               // [m][10]
    int* sf = reinterpret_cast<int*>(hb + (size_t)10 * m);   // [m]
    const double R[2][2] = {{R0, R2}, {R1, R3}};

    if (tid == 0) status[0] = 0;
    // innovation and Jacobian blocks (ekf.jl:55-61)
    const double xv = (double)x[0], yv = (double)x[1], phi = (double)x[2];
    for (int i = tid; i < m; i += nt) {
        const int f = 3 + 2 * (idf[i] - 1);
        const ObsModel om = obs_model(xv, yv, phi, (double)x[f], (double)x[f + 1]);
        for (int q = 0; q < 6; ++q) hb[10 * i + q] = om.Hv[q];
        for (int q = 0; q < 4; ++q) hb[10 * i + 6 + q] = om.Hf[q];
        sf[i] = f;
        vvec[2 * i] = z[2 * i] - om.zp[0];
        vvec[2 * i + 1] = mpi_to_pi_d(z[2 * i + 1] - om.zp[1]);
    }
    for (int a = k + tid; a < kp; a += nt) vvec[a] = 0.0;
    __syncthreads();

    // S = H*PHt + RR (ekf.jl:68); rows/cols >= k are padded with the identity
    for (int idx = tid; idx < kp * kp; idx += nt) {
        const int a = idx / kp, b = idx - a * kp;
        double s;
        if (a < k && b < k) {
            const int i = a >> 1, ra = a & 1;
            const double* h = hb + 10 * i;
            const int f = sf[i];
            s = h[3 * ra + 0] * (double)PHt[(size_t)0 * pht_pitch + b] +
                h[3 * ra + 1] * (double)PHt[(size_t)1 * pht_pitch + b] +
                h[3 * ra + 2] * (double)PHt[(size_t)2 * pht_pitch + b] +
                h[6 + 2 * ra + 0] * (double)PHt[(size_t)f * pht_pitch + b] +
                h[6 + 2 * ra + 1] * (double)PHt[(size_t)(f + 1) * pht_pitch + b];
            if ((b >> 1) == i) s += R[ra][b & 1];
        } else {
            s = (a == b) ? 1.0 : 0.0;
        }
        M[(size_t)a * mp + b] = s;
    }
    __syncthreads();
    // S = (S + S')*0.5 (ekf.jl:69)
    for (int idx = tid; idx < k * k; idx += nt) {
        const int a = idx / k, b = idx - a * k;
        if (a < b) {
            const double s = (M[(size_t)a * mp + b] + M[(size_t)b * mp + a]) * 0.5;
            M[(size_t)a * mp + b] = s;
            M[(size_t)b * mp + a] = s;
        }
    }
    __syncthreads();
    if (Sout) {
        for (int idx = tid; idx < kp * kp; idx += nt) {
            const int a = idx / kp, b = idx - a * kp;
            Sout[(size_t)a * kp + b] = (a < k && b < k) ? M[(size_t)a * mp + b] : 0.0;
        }
    }

    // elimination (see header comment)
    bool bad = false;
    for (int j = 0; j < k; ++j) {
        const double piv = M[(size_t)j * mp + j];
        if (!(piv > 0.0) || piv == __builtin_inf()) { bad = true; break; }   // uniform: every thread reads the same word
        const double rp = 1.0 / piv;
        for (int i = j + 1 + tid; i < k; i += nt) mvec[i] = M[(size_t)i * mp + j] * rp;
        __syncthreads();
        const int rows = k - j - 1;
        for (int idx = tid; idx < rows * k; idx += nt) {
            const int ii = idx / k;
            const int c = idx - ii * k;
            const int i = j + 1 + ii;
            const double mi = mvec[i];
            if (c == j) M[(size_t)i * mp + j] = -mi;
            else M[(size_t)i * mp + c] -= mi * M[(size_t)j * mp + c];
        }
        __syncthreads();
    }
    if (bad) {
        if (tid == 0) { status[0] = 1; status[1] = 1; }    // [1] is sticky until slam_ekf_sync reads it
        return;
    }
    // mvec <- 1/sqrt(D)
    for (int b = tid; b < kp; b += nt) mvec[b] = (b < k) ? 1.0 / sqrt(M[(size_t)b * mp + b]) : 0.0;
    __syncthreads();
    // y = C'*v :  y[b] = (v[b] + sum_{a<b} Linv[b][a] v[a]) / sqrt(D_b)
    for (int b = tid; b < kp; b += nt) {
        double s = 0.0;
        if (b < k) {
            s = vvec[b];
            for (int a = 0; a < b; ++a) s += M[(size_t)b * mp + a] * vvec[a];
            s *= mvec[b];
        }
        yvec[b] = s;
    }
    __syncthreads();
    // g = C*y = inv(S)*v  (x += PHt*g  ==  x += W*v, ekf.jl:72,74)
    for (int a = tid; a < kp; a += nt) {
        double s = 0.0;
        if (a < k) {
            s = yvec[a] * mvec[a];
            for (int b = a + 1; b < k; ++b) s += M[(size_t)b * mp + a] * mvec[b] * yvec[b];
        }
        gvec[a] = s;
    }
    if (!want_sinv) {
        // C[a][b] = Linv[b][a]/sqrt(D_b) (a<b), 1/sqrt(D_b) (a==b), 0 below and in the padding
        for (int idx = tid; idx < kp * kp; idx += nt) {
            const int a = idx / kp, b = idx - a * kp;
            double c = 0.0;
            if (a < k && b < k) {
                if (a == b) c = mvec[b];
                else if (a < b) c = M[(size_t)b * mp + a] * mvec[b];
            }
            Cout[(size_t)a * c_pitch + b] = (T)c;
        }
    } else {
        // inv(S) = C*C' :  Sinv[a][b] = sum_{c >= max(a,b)} C[a][c] C[b][c]   (Joseph form needs K = PHt*inv(S))
        for (int idx = tid; idx < kp * kp; idx += nt) {
            const int a = idx / kp, b = idx - a * kp;
            double s = 0.0;
            if (a < k && b < k) {
                const int c0 = a > b ? a : b;
                for (int c = c0; c < k; ++c) {
                    const double ca = (c == a) ? 1.0 : M[(size_t)c * mp + a];
                    const double cb = (c == b) ? 1.0 : M[(size_t)c * mp + b];
                    s += ca * cb * mvec[c] * mvec[c];
                }
            }
            Cout[(size_t)a * c_pitch + b] = (T)s;
        }
    }
}

// ---------------------------------------------------------------------------
// K5: OUT = beta*ADD + alpha*(IN * MAT)   (n x kp) = (n x kp)(kp x kp)
// ROWS=64 rows per block, 32 output columns per block (blockIdx.y), inner
// dimension staged through LDS in chunks of 128.  `upper` skips the part of an
// upper-triangular MAT that is structurally zero.
// ---------------------------------------------------------------------------
constexpr int PG_ROWS = 64;
constexpr int PG_COLS = 32;
template <typename T> struct PgKc { static constexpr int value = 128; };
template <> struct PgKc<double> { static constexpr int value = 64; };

template <typename T>
__global__ __launch_bounds__(256) void panel_gemm_kernel(const T* __restrict__ IN, int in_pitch, const T* __restrict__ MAT,
                                                          int mat_pitch, int kp, int upper, T alpha, const T* __restrict__ ADD,
                                                          int add_pitch, T beta, T* __restrict__ OUT1, int out1_pitch,
                                                          int out1_col, T* __restrict__ OUT2, int out2_pitch, int out2_col,
                                                          const int32_t* __restrict__ status) {
    if (status[0] != 0) return;
    constexpr int PG_KC = PgKc<T>::value;
    __shared__ T sIn[PG_ROWS][PG_KC + 1];
    __shared__ __attribute__((aligned(16))) T sMat[PG_KC][PG_COLS];
    const int tid = threadIdx.x;
    const int r = tid & 63;
    const int cq = tid >> 6;
    const int row0 = blockIdx.x * PG_ROWS;
    const int b0 = blockIdx.y * PG_COLS;
    T acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = (T)0;
    const int a_end = upper ? (b0 + PG_COLS < kp ? b0 + PG_COLS : kp) : kp;
    for (int a0 = 0; a0 < a_end; a0 += PG_KC) {
        const int ac = (a_end - a0 < PG_KC) ? a_end - a0 : PG_KC;     // multiple of 32
        // IN tile: 64 rows x ac columns, coalesced along the row
        for (int idx = tid; idx < PG_ROWS * ac; idx += 256) {
            const int rr = idx / ac, cc = idx - rr * ac;
            sIn[rr][cc] = IN[(size_t)(row0 + rr) * in_pitch + a0 + cc];
        }
        for (int idx = tid; idx < ac * PG_COLS; idx += 256) {
            const int aa = idx / PG_COLS, bb = idx - aa * PG_COLS;
            sMat[aa][bb] = MAT[(size_t)(a0 + aa) * mat_pitch + b0 + bb];
        }
        __syncthreads();
        for (int a = 0; a < ac; ++a) {
            const T p = sIn[r][a];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] += p * sMat[a][cq * 8 + q];
        }
        __syncthreads();
    }
    const size_t row = (size_t)(row0 + r);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int b = b0 + cq * 8 + q;
        T v = alpha * acc[q];
        if (ADD) v += beta * ADD[row * add_pitch + b];
        OUT1[row * out1_pitch + out1_col + b] = v;
        if (OUT2) OUT2[row * out2_pitch + out2_col + b] = v;
    }
}

// x += PHt * g      (ekf.jl:74 with W*v = PHt*(C*C'*v))
template <typename T>
__global__ __launch_bounds__(256) void x_update_kernel(T* __restrict__ x, const T* __restrict__ PHt, int pitch, int n, int k,
                                                        const double* __restrict__ g, const int32_t* __restrict__ status) {
    if (status[0] != 0) return;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const T* row = PHt + (size_t)r * pitch;
    double s = 0.0;
    for (int a = 0; a < k; ++a) s += (double)row[a] * g[a];
    x[r] = (T)((double)x[r] + s);
}

template <typename T>
int update_typed(slam_ekf* h, int m, const double R[4], int form) {
    const int n = 3 + 2 * h->N;
    const int k = 2 * m;
    const int kp = round_up(k, SLAM_KPAD);
    const int pitchA = h->kcap;         // PHt, Cmat
    const int pitchW = 2 * h->kcap;     // W1, W2
    T* x = (T*)h->x;
    T* P = (T*)h->P;
    T* PHt = (T*)h->PHt;
    T* W1 = (T*)h->W1;
    T* W2 = (T*)h->W2;
    T* Cm = (T*)h->Cmat;

    {   // K2/K3
        KTimer t(h, SLAM_K_PHT);
        const size_t shm = (size_t)m * (10 * sizeof(double) + sizeof(int));
        hipLaunchKernelGGL(pht_kernel<T>, dim3((n + 255) / 256), dim3(256), shm, h->stream, x, P, h->ld, n, h->idfbuf, m, k,
                           kp, PHt, pitchA);
    }
    HIP_TRY(hipGetLastError());
    {   // K4
        KTimer t(h, SLAM_K_FACTOR);
        const bool in_lds = kp <= 128;
        const size_t aux = (size_t)3 * kp * sizeof(double) + (size_t)m * (10 * sizeof(double) + sizeof(int));
        const size_t shm = aux + (in_lds ? (size_t)kp * (kp + 1) * sizeof(double) : 0);
        hipLaunchKernelGGL(factor_kernel<T>, dim3(1), dim3(FACTOR_THREADS), shm, h->stream, x, PHt, pitchA, h->obsbuf,
                           h->idfbuf, m, k, kp, R[0], R[1], R[2], R[3], Cm, pitchA, h->gvec,
                           form == SLAM_FORM_JOSEPH ? h->Smat : (double*)nullptr, form == SLAM_FORM_JOSEPH ? 1 : 0,
                           in_lds ? (double*)nullptr : h->Mwork, h->d_status);
    }
    HIP_TRY(hipGetLastError());
    const dim3 pg_grid(h->npad / PG_ROWS, kp / PG_COLS);
    int kp_total;
    {   // K5
        KTimer t(h, SLAM_K_W1);
        if (form == SLAM_FORM_CHOLESKY) {
            // W1 = PHt*C
            hipLaunchKernelGGL(panel_gemm_kernel<T>, pg_grid, dim3(256), 0, h->stream, PHt, pitchA, Cm, pitchA, kp, 1, (T)1,
                               (const T*)nullptr, 0, (T)0, W1, pitchW, 0, (T*)nullptr, 0, 0, h->d_status);
            kp_total = kp;
        } else {
            // K = PHt*inv(S)  ->  W1[:, 0:kp] and W2[:, kp:2kp]
            hipLaunchKernelGGL(panel_gemm_kernel<T>, pg_grid, dim3(256), 0, h->stream, PHt, pitchA, Cm, pitchA, kp, 0, (T)1,
                               (const T*)nullptr, 0, (T)0, W1, pitchW, 0, W2, pitchW, kp, h->d_status);
            // S (double) -> dtype copy into Cmat is done by convert below; T = PHt - 0.5*K*S -> W1[:, kp:2kp], W2[:, 0:kp]
            kp_total = 2 * kp;
        }
        hipLaunchKernelGGL(x_update_kernel<T>, dim3((n + 255) / 256), dim3(256), 0, h->stream, x, PHt, pitchA, n, k,
                           h->gvec, h->d_status);
    }
    HIP_TRY(hipGetLastError());
    if (form == SLAM_FORM_JOSEPH) {
        const int rc = joseph_T_pass(h, kp);
        if (rc != SLAM_OK) return rc;
    }
    return launch_downdate(h, kp_total, W1, form == SLAM_FORM_CHOLESKY ? (const void*)W1 : (const void*)W2, pitchW);
}

// dtype copy of the k x k double matrix S into Cmat (row-major, pitch kcap)
template <typename T>
__global__ void convert_s_kernel(const double* __restrict__ S, int kp, T* __restrict__ out, int pitch,
                                 const int32_t* __restrict__ status) {
    if (status[0] != 0) return;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= kp * kp) return;
    const int a = idx / kp, b = idx - a * kp;
    out[(size_t)a * pitch + b] = (T)S[idx];
}

template <typename T>
int joseph_T_typed(slam_ekf* h, int kp) {
    const int pitchA = h->kcap, pitchW = 2 * h->kcap;
    T* PHt = (T*)h->PHt;
    T* W1 = (T*)h->W1;
    T* W2 = (T*)h->W2;
    T* Cm = (T*)h->Cmat;
    KTimer t(h, SLAM_K_W1);
    hipLaunchKernelGGL(convert_s_kernel<T>, dim3((kp * kp + 255) / 256), dim3(256), 0, h->stream, h->Smat, kp, Cm, pitchA,
                       h->d_status);
    const dim3 pg_grid(h->npad / PG_ROWS, kp / PG_COLS);
    // T = PHt - 0.5 * K * S ;  K is W1[:, 0:kp]
    hipLaunchKernelGGL(panel_gemm_kernel<T>, pg_grid, dim3(256), 0, h->stream, (const T*)W1, pitchW, (const T*)Cm, pitchA, kp,
                       0, (T)-0.5, (const T*)PHt, pitchA, (T)1, W1, pitchW, kp, W2, pitchW, 0, h->d_status);
    HIP_TRY(hipGetLastError());
    return SLAM_OK;
}

}  // namespace

int joseph_T_pass(slam_ekf* h, int kp) {
    return h->dtype == SLAM_F32 ? joseph_T_typed<float>(h, kp) : joseph_T_typed<double>(h, kp);
}

int launch_update(slam_ekf* h, int m, const double R[4], int form) {
    return h->dtype == SLAM_F32 ? update_typed<float>(h, m, R, form) : update_typed<double>(h, m, R, form);
}

int update_kernels_init() {
    // the factor kernel keeps a 128 x 129 double matrix in LDS: raise the dynamic-LDS cap
    const int big = 160 * 1024;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&factor_kernel<float>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, big));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&factor_kernel<double>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, big));
    return SLAM_OK;
}
