// Focus control of the image context ("adjust_rank", reference app.py:48-127) on the device, fp32 throughout.
//
// Reference, per sample x [L, C] (L = 256 local CLIP tokens, C = 768), in fp32:
//     std_save = x.std()                                  (unbiased, over all L*C values)
//     A        = x - x.mean(-1, keepdim=True)             (row means removed)
//     u, s, v  = torch.pca_lowrank(A, q=20, center=False, niter=100)
//     s[i]    *= f_i(lvl)                                 (i = 0..1 for lvl < 0.5; i = 5..19 and remainder dropped for lvl > 0.5)
//     x_new    = u diag(s) v^T + row means (+ A - u diag(s0) v^T)       then  x_new * std_save / x_new.std()
// pca_lowrank with 100 power iterations converges on the top-q singular triplets, and s_i u_i v_i^T = u_i (u_i^T A), so
//     x_new = keep * A + sum_i g_i u_i (u_i^T A) + row means,   g_i = f_i - keep,  keep = 1 (lvl < 0.5) or 0 (lvl > 0.5)
// needs only the top-q LEFT singular vectors = top eigenvectors of the Gram matrix G = A A^T (L x L): no singular values,
// no V.  They come from a blocked subspace iteration on G (32 columns, re-orthonormalised with Cholesky-QR twice per
// iteration, Rayleigh-Ritz + a 32x32 Jacobi eigen-solve at the end), all inside one workgroup per sample.
//
// Kernels: centre (row means, A, sum / sum-of-squares of x) -> Gram -> eigenvectors -> Z = U^T A -> combine (+ statistics
// of x_new) -> rescale to fp16.  Everything is a few hundred microseconds of a once-per-input-image step.
#include "vd_common.h"
#include "../../include/vd_hip.h"

namespace {

constexpr int P = 32;        // subspace columns (q <= 32)
constexpr int PS = P + 1;    // padded row stride of the LDS matrices

// ---- row means, centred matrix, statistics of x ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void ar_center_kernel(const f16* x, float* A, float* mean, float* rowstat, int L, int C) {
    __shared__ float red[3][256];
    const int l = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const f16* xr = x + ((size_t)b * L + l) * C;
    float s = 0.f, q = 0.f;
    for (int c = tid; c < C; c += 256) {
        const float v = (float)xr[c];
        s += v;
        q += v * v;
    }
    red[0][tid] = s;
    red[1][tid] = q;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { red[0][tid] += red[0][tid + o]; red[1][tid] += red[1][tid + o]; }
        __syncthreads();
    }
    const float m = red[0][0] / (float)C;
    float* ar = A + ((size_t)b * L + l) * C;
    for (int c = tid; c < C; c += 256) ar[c] = (float)xr[c] - m;
    if (tid == 0) {
        mean[(size_t)b * L + l] = m;
        rowstat[((size_t)b * L + l) * 2] = red[0][0];
        rowstat[((size_t)b * L + l) * 2 + 1] = red[1][0];
    }
}

// ---- G = A A^T (L x L), 16 x 16 tiles --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ar_gram_kernel(const float* A, float* G, int L, int C) {
    __shared__ float ta[16][17], tb[16][17];
    const int b = blockIdx.z, tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i = blockIdx.y * 16 + ty, j = blockIdx.x * 16 + tx;
    const float* Ab = A + (size_t)b * L * C;
    float acc = 0.f;
    for (int k0 = 0; k0 < C; k0 += 16) {
        const int ra = blockIdx.y * 16 + ty, rb = blockIdx.x * 16 + ty;
        ta[ty][tx] = (ra < L && k0 + tx < C) ? Ab[(size_t)ra * C + k0 + tx] : 0.f;
        tb[ty][tx] = (rb < L && k0 + tx < C) ? Ab[(size_t)rb * C + k0 + tx] : 0.f;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) acc += ta[ty][k] * tb[tx][k];
        __syncthreads();
    }
    if (i < L && j < L) G[((size_t)b * L + i) * L + j] = acc;
}

// ---- top eigenvectors of G by subspace iteration; one workgroup of 1024 threads per sample ------------------------------
// thread t: rows t/4 + 256*k, columns 8*(t%4) .. +7 of the L x P iterate
__device__ __forceinline__ void chol_qr(float* Y, float* S, int L, int tid) {
    // S = Y^T Y
    {
        const int i = tid >> 5, j = tid & 31;
        float acc = 0.f;
        for (int r = 0; r < L; ++r) acc += Y[r * PS + i] * Y[r * PS + j];
        S[i * PS + j] = acc;
    }
    __syncthreads();
    // Cholesky S = R^T R in place (upper triangle holds R); right-looking, one column per step
    for (int k = 0; k < P; ++k) {
        if (tid == 0) S[k * PS + k] = sqrtf(fmaxf(S[k * PS + k], 1e-30f));
        __syncthreads();
        if (tid > k && tid < P) S[k * PS + tid] /= S[k * PS + k];
        __syncthreads();
        {
            const int i = tid >> 5, j = tid & 31;
            if (i > k && j >= i) S[i * PS + j] -= S[k * PS + i] * S[k * PS + j];
        }
        __syncthreads();
    }
    // Y <- Y R^-1 (forward substitution along each row); one thread per row
    for (int r = tid; r < L; r += 1024) {
        float* y = Y + r * PS;
        for (int j = 0; j < P; ++j) {
            float v = y[j];
            for (int i = 0; i < j; ++i) v -= y[i] * S[i * PS + j];
            y[j] = v / S[j * PS + j];
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void mul_g(const float* Gb, const float* Q, float* Y, int L, int tid) {
    const int cq = (tid & 3) * 8;
    for (int r = tid >> 2; r < L; r += 256) {
        float acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = 0.f;
        for (int k = 0; k < L; ++k) {
            const float g = Gb[(size_t)k * L + r];  // G is symmetric: column r read as row k -> coalesced over r
            const float* qk = Q + k * PS + cq;
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[c] += g * qk[c];
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) Y[r * PS + cq + c] = acc[c];
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void ar_eig_kernel(const float* G, float* U, int L, int q, int iters) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Q = sm;                  // [L][PS]
    float* Y = Q + (size_t)L * PS;  // [L][PS]
    float* S = Y + (size_t)L * PS;  // [P][PS]  Gram / Cholesky factor / Rayleigh quotient T
    float* W = S + P * PS;          // [P][PS]  eigenvectors of T
    float* ev = W + P * PS;         // [P] eigenvalues, then [P] order
    __shared__ float cs[16][2];
    __shared__ int pr[16][2];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* Gb = G + (size_t)b * L * L;
    // deterministic start: hashed pseudo-random entries (any start with a component in every top eigenvector works)
    for (int e = tid; e < L * P; e += 1024) {
        unsigned h = (unsigned)e * 2654435761u + 12345u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        Y[(e / P) * PS + (e % P)] = (float)(h & 0xffffff) / 8388608.0f - 1.0f;
    }
    __syncthreads();
    chol_qr(Y, S, L, tid);
    for (int it = 0; it < iters; ++it) {
        for (int e = tid; e < L * P; e += 1024) Q[(e / P) * PS + (e % P)] = Y[(e / P) * PS + (e % P)];
        __syncthreads();
        mul_g(Gb, Q, Y, L, tid);
        chol_qr(Y, S, L, tid);
        chol_qr(Y, S, L, tid);  // second pass: Cholesky-QR alone loses orthogonality like cond(Y)^2
    }
    // Rayleigh-Ritz: Q = orthonormal basis, T = Q^T G Q
    for (int e = tid; e < L * P; e += 1024) Q[(e / P) * PS + (e % P)] = Y[(e / P) * PS + (e % P)];
    __syncthreads();
    mul_g(Gb, Q, Y, L, tid);
    {
        const int i = tid >> 5, j = tid & 31;
        float acc = 0.f;
        for (int r = 0; r < L; ++r) acc += Q[r * PS + i] * Y[r * PS + j];
        S[i * PS + j] = acc;
        W[i * PS + j] = (i == j) ? 1.f : 0.f;
    }
    __syncthreads();
    {   // symmetrise
        const int i = tid >> 5, j = tid & 31;
        const float t = 0.5f * (S[i * PS + j] + S[j * PS + i]);
        __syncthreads();
        S[i * PS + j] = t;
        __syncthreads();
    }
    // parallel cyclic Jacobi on the 32 x 32 matrix: round-robin pairing, 16 disjoint rotations per round
    for (int sweep = 0; sweep < 12; ++sweep) {
        for (int round = 0; round < P - 1; ++round) {
            if (tid < 16) {
                // tournament schedule: player 31 fixed, the others rotate
                int a = (tid == 0) ? P - 1 : (round + tid) % (P - 1);
                int c = (round + P - 1 - tid) % (P - 1);
                if (tid == 0) c = round % (P - 1);
                const int p = a < c ? a : c, r = a < c ? c : a;
                pr[tid][0] = p;
                pr[tid][1] = r;
                const float app = S[p * PS + p], arr = S[r * PS + r], apr = S[p * PS + r];
                float cc = 1.f, ss = 0.f;
                if (fabsf(apr) > 1e-30f * (fabsf(app) + fabsf(arr)) + 1e-38f) {
                    const float tau = (arr - app) / (2.f * apr);
                    const float t = (tau >= 0.f ? 1.f : -1.f) / (fabsf(tau) + sqrtf(1.f + tau * tau));
                    cc = 1.f / sqrtf(1.f + t * t);
                    ss = t * cc;
                }
                cs[tid][0] = cc;
                cs[tid][1] = ss;
            }
            __syncthreads();
            // rows: S <- J^T S ; thread (pair, column)
            if (tid < 16 * P) {
                const int k = tid >> 5, j = tid & 31;
                const int p = pr[k][0], r = pr[k][1];
                const float c = cs[k][0], s = cs[k][1];
                const float sp = S[p * PS + j], sr = S[r * PS + j];
                S[p * PS + j] = c * sp - s * sr;
                S[r * PS + j] = s * sp + c * sr;
            }
            __syncthreads();
            // columns: S <- S J, W <- W J ; thread (pair, row)
            if (tid < 16 * P) {
                const int k = tid >> 5, i = tid & 31;
                const int p = pr[k][0], r = pr[k][1];
                const float c = cs[k][0], s = cs[k][1];
                const float sp = S[i * PS + p], sr = S[i * PS + r];
                S[i * PS + p] = c * sp - s * sr;
                S[i * PS + r] = s * sp + c * sr;
                const float wp = W[i * PS + p], wr = W[i * PS + r];
                W[i * PS + p] = c * wp - s * wr;
                W[i * PS + r] = s * wp + c * wr;
            }
            __syncthreads();
        }
    }
    if (tid < P) ev[tid] = S[tid * PS + tid];
    __syncthreads();
    if (tid < P) {  // rank of each eigenvalue in descending order (ties broken by index)
        int rank = 0;
        const float v = ev[tid];
        for (int k = 0; k < P; ++k) rank += (ev[k] > v || (ev[k] == v && k < tid)) ? 1 : 0;
        ev[P + rank] = (float)tid;
    }
    __syncthreads();
    // U[:, i] = Q W[:, order[i]], i < q
    for (int e = tid; e < L * q; e += 1024) {
        const int r = e / q, i = e - r * q;
        const int col = (int)ev[P + i];
        float acc = 0.f;
        for (int k = 0; k < P; ++k) acc += Q[r * PS + k] * W[k * PS + col];
        U[((size_t)b * L + r) * q + i] = acc;
    }
}

// ---- Z = U^T A  (q x C) ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ar_z_kernel(const float* A, const float* U, float* Z, int L, int C, int q) {
    const int b = blockIdx.z, i = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float* Ab = A + (size_t)b * L * C;
    const float* Ub = U + (size_t)b * L * q;
    float acc = 0.f;
    for (int r = 0; r < L; ++r) acc += Ub[(size_t)r * q + i] * Ab[(size_t)r * C + c];
    Z[((size_t)b * q + i) * C + c] = acc;
}

// ---- x_new = keep * A + sum_i g_i U[:, i] Z[i, :] + row mean ; per-row statistics of x_new -----------------------------
__global__ __launch_bounds__(256) void ar_combine_kernel(float* A, const float* U, const float* Z, const float* mean, const float* g,
                                                         float keep, float* rowstat2, int L, int C, int q) {
    __shared__ float red[2][256];
    __shared__ float coef[P];
    const int l = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    if (tid < q) coef[tid] = g[tid] * U[((size_t)b * L + l) * q + tid];
    __syncthreads();
    float* ar = A + ((size_t)b * L + l) * C;
    const float m = mean[(size_t)b * L + l];
    float s = 0.f, sq = 0.f;
    for (int c = tid; c < C; c += 256) {
        float v = keep * ar[c] + m;
        for (int i = 0; i < q; ++i) v += coef[i] * Z[((size_t)b * q + i) * C + c];
        ar[c] = v;
        s += v;
        sq += v * v;
    }
    red[0][tid] = s;
    red[1][tid] = sq;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { red[0][tid] += red[0][tid + o]; red[1][tid] += red[1][tid + o]; }
        __syncthreads();
    }
    if (tid == 0) {
        rowstat2[((size_t)b * L + l) * 2] = red[0][0];
        rowstat2[((size_t)b * L + l) * 2 + 1] = red[1][0];
    }
}

// ---- y = x_new * std(x) / std(x_new)  (unbiased, over all L*C values) -> fp16 ------------------------------------------
__global__ __launch_bounds__(256) void ar_finalize_kernel(const float* Xn, const float* rowstat, const float* rowstat2, f16* y, int L, int C) {
    __shared__ double red[4][256];
    const int l = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    double a[4] = {0, 0, 0, 0};
    for (int r = tid; r < L; r += 256) {
        a[0] += rowstat[((size_t)b * L + r) * 2];
        a[1] += rowstat[((size_t)b * L + r) * 2 + 1];
        a[2] += rowstat2[((size_t)b * L + r) * 2];
        a[3] += rowstat2[((size_t)b * L + r) * 2 + 1];
    }
    for (int k = 0; k < 4; ++k) red[k][tid] = a[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o)
            for (int k = 0; k < 4; ++k) red[k][tid] += red[k][tid + o];
        __syncthreads();
    }
    const double n = (double)L * (double)C;
    const double var0 = (red[1][0] - red[0][0] * red[0][0] / n) / (n - 1.0);
    const double var1 = (red[3][0] - red[2][0] * red[2][0] / n) / (n - 1.0);
    const float scale = (float)sqrt(fmax(var0, 0.0) / fmax(var1, 1e-300));
    const float* xr = Xn + ((size_t)b * L + l) * C;
    f16* yr = y + ((size_t)b * L + l) * C;
    for (int c = tid; c < C; c += 256) yr[c] = (f16)(xr[c] * scale);
}

}  // namespace

extern "C" size_t vd_adjust_rank_workspace_bytes(int B, int L, int C, int q) {
    if (B <= 0 || L <= 0 || C <= 0 || q <= 0) return 0;
    const size_t per = (size_t)L * C + (size_t)L * L + (size_t)L * q + (size_t)q * C + (size_t)L * 5 + 64;
    return (size_t)B * per * sizeof(float);
}

extern "C" int vd_adjust_rank_f16(const void* x, void* y, int B, int L, int C, int q, const float* g, float keep, int iters,
                                  float* ws, hipStream_t stream) {
    VD_REQUIRE(x && y && g && ws, "vd_adjust_rank_f16: null pointer");
    VD_REQUIRE(B > 0 && L >= P && L <= 512 && C > 1, "vd_adjust_rank_f16: need 32 <= L <= 512 tokens (L=%d)", L);
    VD_REQUIRE(q > 0 && q <= P, "vd_adjust_rank_f16: rank q=%d must be in 1..%d", q, P);
    VD_REQUIRE(iters > 0 && iters <= 1000, "vd_adjust_rank_f16: iters=%d", iters);
    float* A = ws;
    float* G = A + (size_t)B * L * C;
    float* U = G + (size_t)B * L * L;
    float* Z = U + (size_t)B * L * q;
    float* mean = Z + (size_t)B * q * C;
    float* rs1 = mean + (size_t)B * L;
    float* rs2 = rs1 + (size_t)B * L * 2;
    hipLaunchKernelGGL(ar_center_kernel, dim3(L, B), dim3(256), 0, stream, (const f16*)x, A, mean, rs1, L, C);
    hipLaunchKernelGGL(ar_gram_kernel, dim3((L + 15) / 16, (L + 15) / 16, B), dim3(256), 0, stream, A, G, L, C);
    const size_t lds = ((size_t)2 * L * PS + 2 * P * PS + 2 * P) * sizeof(float);
    static bool attr_done = false;  // idempotent; a race only repeats the call
    if (!attr_done) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ar_eig_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)(((size_t)2 * 512 * PS + 2 * P * PS + 2 * P) * sizeof(float)));
        if (e != hipSuccess) {
            vd_set_error("vd_adjust_rank_f16: cannot reserve LDS: %s", hipGetErrorString(e));
            return VD_ERR_LAUNCH;
        }
        attr_done = true;
    }
    hipLaunchKernelGGL(ar_eig_kernel, dim3(B), dim3(1024), lds, stream, G, U, L, q, iters);
    hipLaunchKernelGGL(ar_z_kernel, dim3((C + 255) / 256, q, B), dim3(256), 0, stream, A, U, Z, L, C, q);
    hipLaunchKernelGGL(ar_combine_kernel, dim3(L, B), dim3(256), 0, stream, A, U, Z, mean, g, keep, rs2, L, C, q);
    hipLaunchKernelGGL(ar_finalize_kernel, dim3(L, B), dim3(256), 0, stream, A, rs1, rs2, (f16*)y, L, C);
    return vd_check_launch("vd_adjust_rank_f16");
}
