// HBM-bound helper kernels of the sampling path (gfx950): timestep embedding, fused CFG + DDIM
// update, q_sample, layout changes at the NCHW API boundary, small-Cin im2col, VAE posterior
// sampling, and the CLIP embedding / pooled-norm helpers.  All math in fp32, storage fp16.
#include "vd_common.h"
#include "../../include/vd_hip.h"
#include <stdarg.h>
#include <stdio.h>

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_vd_err[512] = "";

void vd_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_vd_err, sizeof(g_vd_err), fmt, ap);
    va_end(ap);
}

int vd_check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        vd_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return VD_ERR_LAUNCH;
    }
    return VD_OK;
}

extern "C" const char* vd_last_error(void) { return g_vd_err; }
extern "C" int vd_abi_version(void) { return VD_HIP_ABI_VERSION; }

namespace {

inline int grid_for(size_t n, int per_block = 256, int cap = 8192) {
    size_t g = (n + per_block - 1) / per_block;
    if (g > (size_t)cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// lib/model_zoo/diffusion_utils.py:131-151 -- freqs = exp(-ln(max_period) * i / half), [cos | sin]
__global__ void timestep_embedding_kernel(const int64_t* t, f16* out, int B, int dim, float log_max_period) {
    const int half = dim / 2;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * half; i += gridDim.x * blockDim.x) {
        const int b = i / half, j = i - b * half;
        const float freq = expf(-log_max_period * (float)j / (float)half);
        const float arg = (float)t[b] * freq;
        out[(size_t)b * dim + j] = (f16)cosf(arg);
        out[(size_t)b * dim + half + j] = (f16)sinf(arg);
        if ((dim & 1) && j == 0) out[(size_t)b * dim + dim - 1] = (f16)0.f;
    }
}

// lib/model_zoo/ddim.py:144-170
__global__ void cfg_ddim_kernel(const f16* x, const f16* eps, const f16* noise, f16* x_prev, f16* pred_x0, size_t n,
                                int guided, float s, float rsqrt_at, float sqrt_aprev, float dir_coef, float sigma,
                                float sqrt_1mat) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float e = (float)eps[i];
        if (guided) {
            const float ec = (float)eps[n + i];
            e = e + s * (ec - e);
        }
        const float xv = (float)x[i];
        const float p0 = (xv - sqrt_1mat * e) * rsqrt_at;
        float xp = sqrt_aprev * p0 + dir_coef * e;
        if (noise != nullptr) xp += sigma * (float)noise[i];
        x_prev[i] = (f16)xp;
        if (pred_x0 != nullptr) pred_x0[i] = (f16)p0;
    }
}

// same update with the six step scalars read from device memory, so ONE captured HIP graph serves all DDIM steps
// coef = {guidance scale, 1/sqrt(a_t), sqrt(a_prev), sqrt(1 - a_prev - sigma^2), sigma, sqrt(1 - a_t)}
__global__ void cfg_ddim_dev_kernel(const f16* x, const f16* eps, const f16* noise, f16* x_prev, f16* pred_x0, size_t n,
                                    int guided, const float* coef) {
    const float s = coef[0], rsqrt_at = coef[1], sqrt_aprev = coef[2], dir_coef = coef[3], sigma = coef[4], sqrt_1mat = coef[5];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float e = (float)eps[i];
        if (guided) {
            const float ec = (float)eps[n + i];
            e = e + s * (ec - e);
        }
        const float p0 = ((float)x[i] - sqrt_1mat * e) * rsqrt_at;
        float xp = sqrt_aprev * p0 + dir_coef * e;
        if (noise != nullptr) xp += sigma * (float)noise[i];
        x_prev[i] = (f16)xp;
        if (pred_x0 != nullptr) pred_x0[i] = (f16)p0;
    }
}

__global__ void q_sample_kernel(const f16* x0, const f16* noise, const float* sa, const float* sb, f16* out, int B,
                                size_t per_batch) {
    const size_t n = (size_t)B * per_batch;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / per_batch);
        out[i] = (f16)(sa[b] * (float)x0[i] + sb[b] * (float)noise[i]);
    }
}

// NCHW -> NHWC through a 32x(32+1) LDS tile per (b, 32 channels, 32 pixels)
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const f16* x, f16* y, int C, int HW) {
    __shared__ f16 tile[32][33];
    const int b = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, pix = p0 + tx;
        tile[j][tx] = (c < C && pix < HW) ? x[((size_t)b * C + c) * HW + pix] : (f16)0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int pix = p0 + j, c = c0 + tx;
        if (c < C && pix < HW) y[((size_t)b * HW + pix) * C + c] = tile[tx][j];
    }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const f16* x, f16* y, int C, int HW, float scale,
                                                           float shift, int clamp01) {
    __shared__ f16 tile[32][33];
    const int b = blockIdx.z, c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int pix = p0 + j, c = c0 + tx;
        tile[j][tx] = (c < C && pix < HW) ? x[((size_t)b * HW + pix) * C + c] : (f16)0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, pix = p0 + tx;
        if (c < C && pix < HW) {
            float v = (float)tile[tx][j] * scale + shift;
            if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
            y[((size_t)b * C + c) * HW + pix] = (f16)v;
        }
    }
}

__global__ void im2col_small_kernel(const f16* x, f16* a, int B, int C, int Hin, int Win, int Hout, int Wout,
                                    int ksize, int stride, int pad, int64_t sb, int64_t sc, int64_t sy, int64_t sx,
                                    int kpad, float in_scale, float in_shift) {
    const size_t total = (size_t)B * Hout * Wout * kpad;
    const int kk = ksize * ksize * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % kpad);
        const size_t m = i / kpad;
        float v = 0.f;
        if (k < kk) {
            const int tap = k / C, c = k - tap * C;
            const int ky = tap / ksize, kx = tap - ky * ksize;
            const int ox = (int)(m % Wout);
            const size_t t2 = m / Wout;
            const int oy = (int)(t2 % Hout);
            const int b = (int)(t2 / Hout);
            const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
            if ((unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Win)
                v = (float)x[b * sb + c * sc + iy * sy + ix * sx] * in_scale + in_shift;
        }
        a[i] = (f16)v;
    }
}

// lib/model_zoo/distributions.py:24-37
__global__ void diag_gaussian_kernel(const f16* mom, const f16* noise, f16* z, int B, int zc, int HW, float scale) {
    const size_t n = (size_t)B * zc * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int pix = (int)(i % HW);
        const size_t t2 = i / HW;
        const int c = (int)(t2 % zc);
        const int b = (int)(t2 / zc);
        const f16* mp = mom + ((size_t)b * HW + pix) * (2 * zc);
        const float mean = (float)mp[c];
        float logvar = (float)mp[zc + c];
        logvar = fminf(fmaxf(logvar, -30.f), 20.f);
        const float nz = noise ? (float)noise[i] : 0.f;
        z[i] = (f16)((mean + expf(0.5f * logvar) * nz) * scale);
    }
}

__global__ void axpby_kernel(const f16* x, const f16* y, f16* out, float a, float b, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        U4H8 xv, yv, o;
        xv.u = reinterpret_cast<const uint4*>(x)[i];
        yv.u = reinterpret_cast<const uint4*>(y)[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) o.e[j] = (f16)(a * (float)xv.e[j] + b * (float)yv.e[j]);
        reinterpret_cast<uint4*>(out)[i] = o.u;
    }
}
__global__ void axpby_tail_kernel(const f16* x, const f16* y, f16* out, float a, float b, size_t start, size_t n) {
    const size_t i = start + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (f16)(a * (float)x[i] + b * (float)y[i]);
}

// out = f(x) element-wise, f = erf-GELU (op 0) or tanh (op 1): BERT's intermediate activation and pooler of the Optimus
// encoder, kept out of the GEMM epilogue's activation switch (a one-off text encode, not worth code in every instance)
__device__ __forceinline__ float unary_op(int op, float v) {
    if (op == 0) return vd_gelu_erf(v);
    return 1.0f - 2.0f / (1.0f + __expf(2.0f * v));
}
__global__ void unary_kernel(const f16* x, f16* out, int op, size_t n8, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        U4H8 xv, o;
        xv.u = reinterpret_cast<const uint4*>(x)[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) o.e[j] = (f16)unary_op(op, (float)xv.e[j]);
        reinterpret_cast<uint4*>(out)[i] = o.u;
    }
    if (blockIdx.x == 0 && n8 * 8 + threadIdx.x < n && threadIdx.x < 8)
        out[n8 * 8 + threadIdx.x] = (f16)unary_op(op, (float)x[n8 * 8 + threadIdx.x]);
}

__global__ void embed_tokens_kernel(const int64_t* ids, const f16* tok, const f16* pos, f16* out, int B, int L, int C) {
    const int C8 = C / 8;
    const size_t n = (size_t)B * L * C8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int cc = (int)(i % C8);
        const size_t bl = i / C8;
        const int l = (int)(bl % L);
        const int64_t id = ids[bl];
        U4H8 a, p, o;
        a.u = *reinterpret_cast<const uint4*>(tok + (size_t)id * C + cc * 8);
        p.u = *reinterpret_cast<const uint4*>(pos + (size_t)l * C + cc * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) o.e[j] = (f16)((float)a.e[j] + (float)p.e[j]);
        *reinterpret_cast<uint4*>(out + bl * C + cc * 8) = o.u;
    }
}

// out[b][0] = class_emb + pos[0]; out[b][1+i] = patches[b][i] + pos[1+i]; optional per-token scale
__global__ void clip_vision_embed_kernel(const f16* patches, const f16* cls, const f16* pos, const float* tscale,
                                         f16* out, int B, int L, int C) {
    const int C8 = C / 8;
    const size_t n = (size_t)B * L * C8;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int cc = (int)(i % C8);
        const size_t bl = i / C8;
        const int l = (int)(bl % L);
        const int b = (int)(bl / L);
        U4H8 a, p, o;
        if (l == 0) a.u = *reinterpret_cast<const uint4*>(cls + cc * 8);
        else a.u = *reinterpret_cast<const uint4*>(patches + ((size_t)b * (L - 1) + (l - 1)) * C + cc * 8);
        p.u = *reinterpret_cast<const uint4*>(pos + (size_t)l * C + cc * 8);
        const float sc = tscale ? tscale[bl] : 1.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.e[j] = (f16)(((float)a.e[j] + (float)p.e[j]) * sc);
        *reinterpret_cast<uint4*>(out + bl * C + cc * 8) = o.u;
    }
}

// pixels NCHW [B][C][H][W] -> A[B*(H/P)*(W/P)][kpad], k = (c*P + py)*P + px (torch conv weight order)
__global__ void patchify_kernel(const f16* px, f16* a, int B, int C, int H, int W, int P, int kpad) {
    const int gh = H / P, gw = W / P;
    const size_t total = (size_t)B * gh * gw * kpad;
    const int kk = C * P * P;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % kpad);
        const size_t m = i / kpad;
        float v = 0.f;
        if (k < kk) {
            const int c = k / (P * P), rem = k - c * P * P;
            const int py = rem / P, pxx = rem - py * P;
            const int gx = (int)(m % gw);
            const size_t t2 = m / gw;
            const int gy = (int)(t2 % gh);
            const int b = (int)(t2 / gh);
            v = (float)px[(((size_t)b * C + c) * H + gy * P + py) * W + gx * P + pxx];
        }
        a[i] = (f16)v;
    }
}

// one block per batch element: norm of the reference row, then scale all rows
__global__ __launch_bounds__(256) void scale_by_row_norm_kernel(f16* z, const f16* ref, const int32_t* pool_idx,
                                                                const float* row_scale, int L, int C) {
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ float red[4];
    const f16* rr = ref ? (ref + (size_t)b * C) : (z + ((size_t)b * L + (pool_idx ? pool_idx[b] : 0)) * C);
    float s = 0.f;
    for (int i = tid; i < C; i += 256) {
        const float v = (float)rr[i];
        s += v * v;
    }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float inv = rsqrtf(red[0] + red[1] + red[2] + red[3]);
    __syncthreads();  // everyone has read the reference row before it is rescaled in place
    for (int i = tid; i < L * C; i += 256) {
        const int l = i / C;
        const float sc = inv * (row_scale ? row_scale[(size_t)b * L + l] : 1.f);
        f16* p = z + (size_t)b * L * C + i;
        *p = (f16)((float)*p * sc);
    }
}

// ---- LDS transpose-read probe: what ds_read_b64_tr_b16 returns for per-lane byte addresses ----
__global__ void probe_tr16_kernel(const int32_t* addr_bytes, int16_t* out) {
    __shared__ __attribute__((aligned(16))) short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) short*)lds;
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(size_t)(base + (unsigned)addr_bytes[threadIdx.x]));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}

// ---- MFMA layout probe: one wave, one-hot operands, reports what the hardware does ----
__global__ void probe_mfma_kernel(int32_t* out_a_k, int32_t* out_c_row, int32_t* out_c_col) {
    const int lane = threadIdx.x;
    // (1) C/D layout: A = one-hot rows (A[i][0] = i+1 for k=0 holder), B[0][j] = 1 -> D[i][j] = i+1 ... we use two runs
    // run 1: D[i][j] = i   (A[i][k0] = i, B[k0][j] = 1) where k0 is held by (hi=0, jj=0) in both operands
    f16x8 a, bb;
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    for (int j = 0; j < 8; ++j) { a[j] = (f16)0.f; bb[j] = (f16)0.f; }
    if (lane < 32) { a[0] = (f16)(float)(lane & 31); bb[0] = (f16)1.f; }
    f32x16 d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bb, c, 0, 0, 0);
    // run 2: D[i][j] = j
    if (lane < 32) { a[0] = (f16)1.f; bb[0] = (f16)(float)(lane & 31); }
    f32x16 d2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bb, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        out_c_row[lane * 16 + r] = (int)d1[r];
        out_c_col[lane * 16 + r] = (int)d2[r];
    }
    // (2) k pairing: A slot (hi_a, ja) one-hot in row 0, B slot (lane-half hb, jb) one-hot in col 0:
    // D[0][0] != 0 iff the two slots address the same k.  Report for each A slot the matching B slot index.
    for (int sa = 0; sa < 16; ++sa) {
        int match = -1;
        for (int sb = 0; sb < 16; ++sb) {
            for (int j = 0; j < 8; ++j) { a[j] = (f16)0.f; bb[j] = (f16)0.f; }
            if ((lane & 31) == 0 && (lane >> 5) == (sa >> 3)) a[sa & 7] = (f16)1.f;
            if ((lane & 31) == 0 && (lane >> 5) == (sb >> 3)) bb[sb & 7] = (f16)1.f;
            f32x16 dd = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bb, c, 0, 0, 0);
            const float v = __shfl(dd[0], 0, 64);
            if (v != 0.f) match = sb;
        }
        if (lane == 0) out_a_k[sa] = match;
    }
}

}  // namespace

extern "C" int vd_timestep_embedding_f16(const int64_t* t, void* out, int B, int dim, float max_period,
                                         hipStream_t stream) {
    VD_REQUIRE(t && out && B > 0 && dim >= 2, "vd_timestep_embedding_f16: bad arguments");
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3(grid_for((size_t)B * (dim / 2))), dim3(256), 0, stream, t,
                       (f16*)out, B, dim, logf(max_period));
    return vd_check_launch("vd_timestep_embedding_f16");
}

extern "C" int vd_cfg_ddim_step_f16(const void* x, const void* eps, const void* noise, void* x_prev, void* pred_x0,
                                    int64_t n, int guided, float guidance_scale, float a_t, float a_prev, float sigma,
                                    float sqrt_one_minus_at, hipStream_t stream) {
    VD_REQUIRE(x && eps && x_prev && n > 0, "vd_cfg_ddim_step_f16: bad arguments");
    VD_REQUIRE(a_t > 0.f && a_prev >= 0.f, "vd_cfg_ddim_step_f16: alphas must be positive");
    float dir2 = 1.f - a_prev - sigma * sigma;
    if (dir2 < 0.f) dir2 = 0.f;
    hipLaunchKernelGGL(cfg_ddim_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, stream, (const f16*)x,
                       (const f16*)eps, (const f16*)noise, (f16*)x_prev, (f16*)pred_x0, (size_t)n, guided,
                       guidance_scale, 1.0f / sqrtf(a_t), sqrtf(a_prev), sqrtf(dir2), sigma, sqrt_one_minus_at);
    return vd_check_launch("vd_cfg_ddim_step_f16");
}

extern "C" int vd_cfg_ddim_step_dev_f16(const void* x, const void* eps, const void* noise, void* x_prev, void* pred_x0,
                                        int64_t n, int guided, const float* coef, hipStream_t stream) {
    VD_REQUIRE(x && eps && x_prev && coef && n > 0, "vd_cfg_ddim_step_dev_f16: bad arguments");
    hipLaunchKernelGGL(cfg_ddim_dev_kernel, dim3(grid_for((size_t)n)), dim3(256), 0, stream, (const f16*)x,
                       (const f16*)eps, (const f16*)noise, (f16*)x_prev, (f16*)pred_x0, (size_t)n, guided, coef);
    return vd_check_launch("vd_cfg_ddim_step_dev_f16");
}

extern "C" int vd_q_sample_f16(const void* x0, const void* noise, const float* sa, const float* sb, void* out, int B,
                               int64_t per_batch, hipStream_t stream) {
    VD_REQUIRE(x0 && noise && sa && sb && out && B > 0 && per_batch > 0, "vd_q_sample_f16: bad arguments");
    hipLaunchKernelGGL(q_sample_kernel, dim3(grid_for((size_t)B * per_batch)), dim3(256), 0, stream, (const f16*)x0,
                       (const f16*)noise, sa, sb, (f16*)out, B, (size_t)per_batch);
    return vd_check_launch("vd_q_sample_f16");
}

extern "C" int vd_nchw_to_nhwc_f16(const void* x, void* y, int B, int C, int H, int W, hipStream_t stream) {
    VD_REQUIRE(x && y && B > 0 && C > 0 && H > 0 && W > 0, "vd_nchw_to_nhwc_f16: bad arguments");
    const int HW = H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((HW + 31) / 32, (C + 31) / 32, B), dim3(256), 0, stream,
                       (const f16*)x, (f16*)y, C, HW);
    return vd_check_launch("vd_nchw_to_nhwc_f16");
}

extern "C" int vd_nhwc_to_nchw_f16(const void* x, void* y, int B, int C, int H, int W, float scale, float shift,
                                   int clamp01, hipStream_t stream) {
    VD_REQUIRE(x && y && B > 0 && C > 0 && H > 0 && W > 0, "vd_nhwc_to_nchw_f16: bad arguments");
    const int HW = H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((HW + 31) / 32, (C + 31) / 32, B), dim3(256), 0, stream,
                       (const f16*)x, (f16*)y, C, HW, scale, shift, clamp01);
    return vd_check_launch("vd_nhwc_to_nchw_f16");
}

extern "C" int vd_im2col_small_f16(const void* x, void* a, int B, int C, int Hin, int Win, int Hout, int Wout,
                                   int ksize, int stride, int pad, int64_t sb, int64_t sc, int64_t sy, int64_t sx,
                                   int kpad, float in_scale, float in_shift, hipStream_t stream) {
    VD_REQUIRE(x && a && B > 0 && C > 0 && Hout > 0 && Wout > 0 && ksize > 0 && stride > 0, "vd_im2col_small_f16: bad arguments");
    VD_REQUIRE(kpad >= ksize * ksize * C && kpad % 64 == 0, "vd_im2col_small_f16: kpad=%d must be a multiple of 64 and >= %d", kpad, ksize * ksize * C);
    const size_t total = (size_t)B * Hout * Wout * kpad;
    hipLaunchKernelGGL(im2col_small_kernel, dim3(grid_for(total, 256, 16384)), dim3(256), 0, stream, (const f16*)x,
                       (f16*)a, B, C, Hin, Win, Hout, Wout, ksize, stride, pad, sb, sc, sy, sx, kpad, in_scale, in_shift);
    return vd_check_launch("vd_im2col_small_f16");
}

extern "C" int vd_diag_gaussian_sample_f16(const void* moments, const void* noise, void* z, int B, int zc, int HW,
                                           float scale, hipStream_t stream) {
    VD_REQUIRE(moments && z && B > 0 && zc > 0 && HW > 0, "vd_diag_gaussian_sample_f16: bad arguments");
    hipLaunchKernelGGL(diag_gaussian_kernel, dim3(grid_for((size_t)B * zc * HW)), dim3(256), 0, stream,
                       (const f16*)moments, (const f16*)noise, (f16*)z, B, zc, HW, scale);
    return vd_check_launch("vd_diag_gaussian_sample_f16");
}

extern "C" int vd_axpby_f16(const void* x, const void* y, void* out, float a, float b, int64_t n, hipStream_t stream) {
    VD_REQUIRE(x && y && out && n > 0, "vd_axpby_f16: bad arguments");
    const size_t n8 = (size_t)n / 8;
    if (n8 > 0)
        hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n8)), dim3(256), 0, stream, (const f16*)x, (const f16*)y,
                           (f16*)out, a, b, n8);
    if ((size_t)n > n8 * 8)
        hipLaunchKernelGGL(axpby_tail_kernel, dim3(1), dim3(8), 0, stream, (const f16*)x, (const f16*)y, (f16*)out, a,
                           b, n8 * 8, (size_t)n);
    return vd_check_launch("vd_axpby_f16");
}

extern "C" int vd_unary_f16(const void* x, void* out, int op, int64_t n, hipStream_t stream) {
    VD_REQUIRE(x && out && n > 0 && (op == VD_UNARY_GELU_ERF || op == VD_UNARY_TANH), "vd_unary_f16: bad arguments");
    const size_t n8 = (size_t)n / 8;
    hipLaunchKernelGGL(unary_kernel, dim3(grid_for(n8 > 0 ? n8 : 1)), dim3(256), 0, stream, (const f16*)x, (f16*)out, op, n8, (size_t)n);
    return vd_check_launch("vd_unary_f16");
}

extern "C" int vd_embed_tokens_f16(const int64_t* ids, const void* tok_emb, const void* pos_emb, void* out, int B,
                                   int L, int C, hipStream_t stream) {
    VD_REQUIRE(ids && tok_emb && pos_emb && out && B > 0 && L > 0 && C > 0 && C % 8 == 0, "vd_embed_tokens_f16: bad arguments");
    hipLaunchKernelGGL(embed_tokens_kernel, dim3(grid_for((size_t)B * L * (C / 8))), dim3(256), 0, stream, ids,
                       (const f16*)tok_emb, (const f16*)pos_emb, (f16*)out, B, L, C);
    return vd_check_launch("vd_embed_tokens_f16");
}

extern "C" int vd_clip_vision_embed_f16(const void* patches, const void* class_emb, const void* pos_emb,
                                        const float* token_scale, void* out, int B, int L, int C, hipStream_t stream) {
    VD_REQUIRE(patches && class_emb && pos_emb && out && B > 0 && L > 1 && C % 8 == 0, "vd_clip_vision_embed_f16: bad arguments");
    hipLaunchKernelGGL(clip_vision_embed_kernel, dim3(grid_for((size_t)B * L * (C / 8))), dim3(256), 0, stream,
                       (const f16*)patches, (const f16*)class_emb, (const f16*)pos_emb, token_scale, (f16*)out, B, L, C);
    return vd_check_launch("vd_clip_vision_embed_f16");
}

extern "C" int vd_patchify_f16(const void* pixels, void* a, int B, int C, int H, int W, int P, int kpad,
                               hipStream_t stream) {
    VD_REQUIRE(pixels && a && B > 0 && C > 0 && P > 0 && H % P == 0 && W % P == 0, "vd_patchify_f16: bad arguments");
    VD_REQUIRE(kpad >= C * P * P && kpad % 64 == 0, "vd_patchify_f16: kpad=%d must be a multiple of 64 and >= %d", kpad, C * P * P);
    const size_t total = (size_t)B * (H / P) * (W / P) * kpad;
    hipLaunchKernelGGL(patchify_kernel, dim3(grid_for(total, 256, 16384)), dim3(256), 0, stream, (const f16*)pixels,
                       (f16*)a, B, C, H, W, P, kpad);
    return vd_check_launch("vd_patchify_f16");
}

extern "C" int vd_scale_by_row_norm_f16(void* z, const void* ref, const int32_t* pool_idx, const float* row_scale,
                                        int B, int L, int C, hipStream_t stream) {
    VD_REQUIRE(z && B > 0 && L > 0 && C > 0, "vd_scale_by_row_norm_f16: bad arguments");
    hipLaunchKernelGGL(scale_by_row_norm_kernel, dim3(B), dim3(256), 0, stream, (f16*)z, (const f16*)ref, pool_idx,
                       row_scale, L, C);
    return vd_check_launch("vd_scale_by_row_norm_f16");
}

namespace {
// which XCD a block runs on (HW_REG_XCC_ID, hwreg 20, bits 3:0) -- the ticketed split of conv3x3_halo_kernel exchanges slabs
// through the XCD's L2 and relies on the dispatcher's round-robin placement (block b of the linearised grid on XCD b % 8)
__global__ void probe_xcc_kernel(int32_t* out) {
    if (threadIdx.x == 0) out[blockIdx.y * gridDim.x + blockIdx.x] = (int32_t)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15);
}
}  // namespace

extern "C" int vd_probe_xcc_ids(int32_t* out, int grid_x, int grid_y, hipStream_t stream) {
    VD_REQUIRE(out && grid_x > 0 && grid_y > 0, "vd_probe_xcc_ids: bad arguments");
    hipLaunchKernelGGL(probe_xcc_kernel, dim3(grid_x, grid_y), dim3(256), 0, stream, out);
    return vd_check_launch("vd_probe_xcc_ids");
}

extern "C" int vd_probe_lds_tr16(const int32_t* addr_bytes, int16_t* out, hipStream_t stream) {
    VD_REQUIRE(addr_bytes && out, "vd_probe_lds_tr16: null pointer");
    hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, stream, addr_bytes, out);
    return vd_check_launch("vd_probe_lds_tr16");
}

extern "C" int vd_probe_mfma_layout(int32_t* out_a_k, int32_t* out_c_row, int32_t* out_c_col, hipStream_t stream) {
    VD_REQUIRE(out_a_k && out_c_row && out_c_col, "vd_probe_mfma_layout: null pointer");
    hipLaunchKernelGGL(probe_mfma_kernel, dim3(1), dim3(64), 0, stream, out_a_k, out_c_row, out_c_col);
    return vd_check_launch("vd_probe_mfma_layout");
}
