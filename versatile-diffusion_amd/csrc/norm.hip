// GroupNorm(+SiLU) and LayerNorm for channels-last fp16 activations (gfx950).
// HBM-bound: every element is read with 16-byte loads, statistics in fp32.
//
// GroupNorm runs as two launches: per-slab partial sums -> [fold partials to mean/rstd per (b, group), then]
// normalise * gamma + beta (+SiLU).  The input may be the channel-concatenation of two tensors
// (skip connections of the UNet up path), which is read in place.
// Reference: GroupNorm32/normalization (lib/model_zoo/diffusion_utils.py:175-191), SiLU in
// ResBlock.in_layers/out_layers (openaimodel.py:196-200,230-237), Normalize eps=1e-6
// (attention.py:76-77, autokl_modules.py:38-39), nn.LayerNorm (attention.py:205-207).
#include "vd_common.h"
#include "../../include/vd_hip.h"

namespace {

constexpr int GN_MAX_POS = 2;  // channel-chunk positions per thread: supports C <= 2*256*8 = 4096
#ifndef VD_GN_U
#define VD_GN_U 4
#endif
constexpr int GN_U = VD_GN_U;  // rows per trip of the slab loops = 16-byte loads in flight per thread

struct GnGeom {
    int C, C8, TC, R, npos, rows_per_chunk, nchunk;
};

inline GnGeom gn_geom(int HW, int C) {
    GnGeom g;
    g.C = C;
    g.C8 = C / 8;
    g.TC = g.C8 < 256 ? g.C8 : 256;
    g.R = 256 / g.TC;
    g.npos = (g.C8 + g.TC - 1) / g.TC;
    int rpc = 16384 / C;
    if (rpc < 1) rpc = 1;
    const int min_rpc = (HW + 255) / 256;
    if (rpc < min_rpc) rpc = min_rpc;
    rpc = ((rpc + g.R - 1) / g.R) * g.R;
    g.rows_per_chunk = rpc;
    g.nchunk = (HW + rpc - 1) / rpc;
    return g;
}

__device__ __forceinline__ const f16* gn_src(const f16* x0, int c0, const f16* x1, int c1, size_t row, int ch) {
    return (ch < c0) ? (x0 + row * c0 + ch) : (x1 + row * c1 + (ch - c0));
}

// Order-independent accumulation in LDS (round 6).  Float atomics add in arrival order, so two runs of one launch could differ in
// the last bits (found on the 4x4 level of a 32x32-latent forward and in the 0-D flow; the 64x64 / 96x96 geometries never take these
// kernels' atomics).  The single-launch kernels below add their per-thread partials as 64-bit INTEGERS instead (integer adds
// commute): a first sweep takes the block's largest |partial| of each quantity with an integer max on the float bits (exact, order-free),
// which fixes a power-of-two scale 2^se with max * 2^se < 2^39 -- 2^13 contributions stay below 2^52, and every partial keeps its
// fp32 mantissa down to 2^-39 of the largest one.
__device__ __forceinline__ int fx_scale_exp(unsigned maxbits) {
    const int e = (int)((maxbits >> 23) & 255) - 127;   // floor(log2(max)); max == 0: -127
    const int se = 38 - e;
    return se > 120 ? 120 : se;
}
__device__ __forceinline__ void fx_add(long long* p, float v, int se) {
    atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__float2ll_rn(ldexpf(v, se)));
}
__device__ __forceinline__ float fx_get(long long v, int se) { return (float)ldexp((double)v, -se); }

// One-pass variance with a per-group SHIFT: sums run over (x - k_g), k_g = the group's first sample (row 0, first channel
// of the group).  Any sample lies within a few sigma of the mean, so sum((x-k)^2)/n - (sum(x-k)/n)^2 no longer cancels
// catastrophically when |mean| >> sigma (a plain E[x^2] - mean^2 in fp32 collapses to var = 0 there, which matters with
// eps = 1e-6 in the VAE / SpatialTransformer norms); torch's two-pass / Welford form has the same robustness.
__device__ __forceinline__ float gn_shift(const f16* x0, int c0, const f16* x1, int c1, size_t row0, int group, int cg) {
    return (float)*gn_src(x0, c0, x1, c1, row0, group * cg);
}

// shifts of the 8 consecutive channels starting at ch (a multiple of 8): with cg >= 8 (or cg == 4) they span at most two
// groups, so two loads serve all 8
__device__ __forceinline__ void gn_shift8(float* kk, const f16* x0, int c0, const f16* x1, int c1, size_t row0, int ch, int cg) {
    if (cg < 8 && cg != 4) {
#pragma unroll
        for (int i = 0; i < 8; ++i) kk[i] = gn_shift(x0, c0, x1, c1, row0, (ch + i) / cg, cg);
        return;
    }
    const int g_lo = ch / cg, g_hi = (ch + 7) / cg;
    const float k_lo = gn_shift(x0, c0, x1, c1, row0, g_lo, cg);
    const float k_hi = (g_hi == g_lo) ? k_lo : gn_shift(x0, c0, x1, c1, row0, g_hi, cg);
    const int split = g_hi * cg - ch;   // first lane of the second group (>= 8 when there is none)
#pragma unroll
    for (int i = 0; i < 8; ++i) kk[i] = (g_hi != g_lo && i >= split) ? k_hi : k_lo;
}

__global__ __launch_bounds__(256) void gn_partial_kernel(const f16* x0, int c0, const f16* x1, int c1, float* part,
                                                         int HW, int groups, GnGeom g) {
    __shared__ float ls[64 * 2];
    const int tid = threadIdx.x, b = blockIdx.y, chunk = blockIdx.x;
    if (tid < groups * 2) ls[tid] = 0.f;
    __syncthreads();
    const int cg = g.C / groups;
    const int tc = tid % g.TC, rl = tid / g.TC;
    if (rl < g.R) {
        const int r0 = chunk * g.rows_per_chunk;
        int r1 = r0 + g.rows_per_chunk;
        if (r1 > HW) r1 = HW;
        for (int pos = 0; pos < g.npos; ++pos) {
            const int cc = tc + pos * g.TC;
            if (cc >= g.C8) break;
            float s[8], q[8], kk[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] = q[i] = 0.f;
            gn_shift8(kk, x0, c0, x1, c1, (size_t)b * HW, cc * 8, cg);
            // 4 rows per trip, all four 16-byte loads issued before any is consumed: with ~2 blocks per CU a single
            // load in flight per thread left HBM at a quarter of its bandwidth (rows past the slab are clamped and
            // weighted 0 rather than branched around, so the loads stay unconditional)
            for (int r = r0 + rl; r < r1; r += GN_U * g.R) {
                U4H8 t[GN_U];
                float wgt[GN_U];
#pragma unroll
                for (int u = 0; u < GN_U; ++u) {
                    const int ru = r + u * g.R;
                    wgt[u] = ru < r1 ? 1.f : 0.f;
                    t[u].u = *reinterpret_cast<const uint4*>(gn_src(x0, c0, x1, c1, (size_t)b * HW + (ru < r1 ? ru : r), cc * 8));
                }
#pragma unroll
                for (int u = 0; u < GN_U; ++u)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float v = ((float)t[u].e[i] - kk[i]) * wgt[u];
                        s[i] += v;
                        q[i] += v * v;
                    }
            }
            // fold the (up to 8) channels into their groups, one LDS atomic per run
            int gcur = (cc * 8) / cg;
            float as = 0.f, aq = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int gi = (cc * 8 + i) / cg;
                if (gi != gcur) {
                    atomicAdd(&ls[gcur * 2], as);
                    atomicAdd(&ls[gcur * 2 + 1], aq);
                    as = aq = 0.f;
                    gcur = gi;
                }
                as += s[i];
                aq += q[i];
            }
            atomicAdd(&ls[gcur * 2], as);
            atomicAdd(&ls[gcur * 2 + 1], aq);
        }
    }
    __syncthreads();
    if (tid < groups * 2) part[((size_t)b * g.nchunk + chunk) * groups * 2 + tid] = ls[tid];
}

// normalise * gamma + beta (+SiLU).  Every block first folds the slab partial sums of its batch element into
// mean / rstd per group (<= 256 slabs x 64 values, L2-resident) -- cheaper than a separate finalize launch.
__global__ __launch_bounds__(256) void gn_apply_kernel(const f16* x0, int c0, const f16* x1, int c1, const f16* gamma,
                                                       const f16* beta, const float* part, f16* y, int HW, int groups,
                                                       int apply_silu, float inv_count, float eps, GnGeom g) {
    const int tid = threadIdx.x, b = blockIdx.y, chunk = blockIdx.x;
    __shared__ float red[256];
    __shared__ float stat[64 * 2];
    {
        const int ne = groups * 2;
        const int e = tid % ne, lanes = 256 / ne, cl = tid / ne;
        float acc = 0.f;
        if (cl < lanes) {
            const float* pp = part + (size_t)b * g.nchunk * ne + e;
            int c = cl;
            for (; c + 7 * lanes < g.nchunk; c += 8 * lanes) {  // 8 independent loads in flight: this is a latency chain
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = pp[(size_t)(c + u * lanes) * ne];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += t[u];
            }
            for (; c < g.nchunk; c += lanes) acc += pp[(size_t)c * ne];
        }
        red[tid] = (cl < lanes) ? acc : 0.f;
        __syncthreads();
        if (tid < ne) {
            float t = 0.f;
            for (int l = 0; l < lanes; ++l) t += red[l * ne + tid];
            red[tid] = t;
        }
        __syncthreads();
        if (tid < groups) {
            const float ms = red[tid * 2] * inv_count;   // mean of (x - k)
            float var = red[tid * 2 + 1] * inv_count - ms * ms;
            if (var < 0.f) var = 0.f;
            stat[tid * 2] = ms + gn_shift(x0, c0, x1, c1, (size_t)b * HW, tid, g.C / groups);
            stat[tid * 2 + 1] = rsqrtf(var + eps);
        }
        __syncthreads();
    }
    const int cg = g.C / groups;
    const int tc = tid % g.TC, rl = tid / g.TC;
    if (rl >= g.R) return;
    const int r0 = chunk * g.rows_per_chunk;
    int r1 = r0 + g.rows_per_chunk;
    if (r1 > HW) r1 = HW;
    float sc[GN_MAX_POS][8], sh[GN_MAX_POS][8];
#pragma unroll
    for (int pos = 0; pos < GN_MAX_POS; ++pos) {
        const int cc = tc + pos * g.TC;
        if (pos < g.npos && cc < g.C8) {
            U4H8 ga, be;
            ga.u = *reinterpret_cast<const uint4*>(gamma + cc * 8);
            be.u = *reinterpret_cast<const uint4*>(beta + cc * 8);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int gi = (cc * 8 + i) / cg;
                const float mean = stat[gi * 2];
                const float rstd = stat[gi * 2 + 1];
                sc[pos][i] = rstd * (float)ga.e[i];
                sh[pos][i] = (float)be.e[i] - mean * sc[pos][i];
            }
        }
    }
#pragma unroll
    for (int pos = 0; pos < GN_MAX_POS; ++pos) {
        const int cc = tc + pos * g.TC;
        if (pos < g.npos && cc < g.C8) {
            for (int r = r0 + rl; r < r1; r += GN_U * g.R) {  // GN_U loads in flight per thread, see gn_partial_kernel (6 and 12 measured equal)
                U4H8 t[GN_U];
#pragma unroll
                for (int u = 0; u < GN_U; ++u) {
                    const int ru = r + u * g.R;
                    t[u].u = *reinterpret_cast<const uint4*>(gn_src(x0, c0, x1, c1, (size_t)b * HW + (ru < r1 ? ru : r), cc * 8));
                }
#pragma unroll
                for (int u = 0; u < GN_U; ++u) {
                    const int ru = r + u * g.R;
                    U4H8 o;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        float v = (float)t[u].e[i] * sc[pos][i] + sh[pos][i];
                        if (apply_silu) v = vd_silu(v);
                        o.e[i] = (f16)v;
                    }
                    if (ru < r1) *reinterpret_cast<uint4*>(y + ((size_t)b * HW + ru) * g.C + cc * 8) = o.u;
                }
            }
        }
    }
}

// ---- single-launch GroupNorm for the low-resolution levels -------------------------------------------------------------
// One block per (batch element, channel slab), slab = lcm(channels per group, 8) channels = whole groups AND whole
// 16-byte chunks.  When the slab of one sample (HW rows x slab channels) fits in the block's registers (<= GN_SLAB_ITEMS
// 16-byte items per thread) the statistics never leave the block: x is read once, y written once, no scratch, one launch
// instead of two (the two-kernel path above costs ~15 us
// there regardless of size: two launches plus the slab-sum round trip).  Covers the 8x8 and 16x16 levels.
constexpr int GN_SLAB_ITEMS = 12;  // measured: 24 rows per thread (32x32 levels, 64-128 blocks) is no faster than two launches

template <int NITEM>
__global__ __launch_bounds__(256) void gn_slab_kernel(const f16* x0, int c0, const f16* x1, int c1, const f16* gamma,
                                                      const f16* beta, f16* y, int HW, int C, int cg, int slab_chunks,
                                                      int apply_silu, float inv_count, float eps) {
    __shared__ long long ls[16 * 2];  // <= 16 groups per slab: fixed-point (sum, sum of squares), see fx_add
    __shared__ unsigned mx[2];
    __shared__ float stat[16 * 2];
    const int tid = threadIdx.x, b = blockIdx.y;
    const int ch0 = blockIdx.x * slab_chunks * 8;  // first channel of the slab
    const int ngrp = slab_chunks * 8 / cg;
    if (tid < 32) ls[tid] = 0;
    if (tid < 2) mx[tid] = 0u;
    __syncthreads();
    // a thread keeps ONE 8-channel chunk of the slab and walks the rows: its per-channel sums stay in registers and
    // reach LDS once (walking items linearly instead costs an LDS atomic pair per item on a handful of addresses)
    const int rp = 256 / slab_chunks;                 // rows per pass
    const int ck = tid % slab_chunks, rl = tid / slab_chunks;
    const bool active = rl < rp;
    const int ch = ch0 + ck * 8;
    U4H8 t[NITEM];
    float sm[8], sq[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) sm[i] = sq[i] = 0.f;
    // the (up to 8) channels of the chunk folded into their groups, one (sum, sum of squares) pair per run of one group
    auto runs = [&](auto&& emit) {
        int gcur = (ck * 8) / cg;
        float as = 0.f, aq = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int gi = (ck * 8 + i) / cg;
            if (gi != gcur) {
                emit(gcur, as, aq);
                as = aq = 0.f;
                gcur = gi;
            }
            as += sm[i];
            aq += sq[i];
        }
        emit(gcur, as, aq);
    };
    if (active) {
        // all loads unconditional and back to back; rows past HW re-read the last row and are weighted 0
        float wgt[NITEM];
#pragma unroll
        for (int u = 0; u < NITEM; ++u) {
            const int r = rl + u * rp;
            wgt[u] = r < HW ? 1.f : 0.f;
            t[u].u = *reinterpret_cast<const uint4*>(gn_src(x0, c0, x1, c1, (size_t)b * HW + (r < HW ? r : HW - 1), ch));
        }
        float kk[8];
        gn_shift8(kk, x0, c0, x1, c1, (size_t)b * HW, ch, cg);
#pragma unroll
        for (int u = 0; u < NITEM; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float v = ((float)t[u].e[i] - kk[i]) * wgt[u];
                sm[i] += v;
                sq[i] += v * v;
            }
        float ma = 0.f, mq = 0.f;
        runs([&](int, float as, float aq) { ma = fmaxf(ma, fabsf(as)); mq = fmaxf(mq, aq); });
        atomicMax(&mx[0], __float_as_uint(ma));
        atomicMax(&mx[1], __float_as_uint(mq));
    }
    __syncthreads();
    const int ses = fx_scale_exp(mx[0]), seq = fx_scale_exp(mx[1]);
    if (active) runs([&](int g, float as, float aq) { fx_add(&ls[g * 2], as, ses); fx_add(&ls[g * 2 + 1], aq, seq); });
    __syncthreads();
    if (tid < ngrp) {
        const float ms = fx_get(ls[tid * 2], ses) * inv_count;
        float var = fx_get(ls[tid * 2 + 1], seq) * inv_count - ms * ms;
        if (var < 0.f) var = 0.f;
        stat[tid * 2] = ms + gn_shift(x0, c0, x1, c1, (size_t)b * HW, ch0 / cg + tid, cg);
        stat[tid * 2 + 1] = rsqrtf(var + eps);
    }
    __syncthreads();
    if (active) {
        U4H8 ga, be;
        ga.u = *reinterpret_cast<const uint4*>(gamma + ch);
        be.u = *reinterpret_cast<const uint4*>(beta + ch);
        float sc[8], sh[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int gi = (ck * 8 + i) / cg;
            sc[i] = stat[gi * 2 + 1] * (float)ga.e[i];
            sh[i] = (float)be.e[i] - stat[gi * 2] * sc[i];
        }
#pragma unroll
        for (int u = 0; u < NITEM; ++u) {
            const int r = rl + u * rp;
            U4H8 o;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float v = (float)t[u].e[i] * sc[i] + sh[i];
                if (apply_silu) v = vd_silu(v);
                o.e[i] = (f16)v;
            }
            if (r < HW) *reinterpret_cast<uint4*>(y + ((size_t)b * HW + r) * C + ch) = o.u;
        }
    }
}

// ---- GroupNorm of the 0-D (text-latent) data flow ---------------------------------------------------------------------
// FCBlock (reference openaimodel.py:2084-2141) normalises the FLATTENED [C, sdim] vector of a sample with 32 groups and
// an affine per flat element.  Here a sample is [S = sdim][C] channels-last (optionally two tensors concatenated on C),
// so a group is a channel range over all S positions -- the statistics of a regular GroupNorm with HW = S -- but gamma /
// beta are indexed [s][c] (re-ordered once at load from the reference's c * sdim + s).  One block per sample: the whole
// sample is <= 10240 elements.
__global__ __launch_bounds__(256) void gn0d_kernel(const f16* x0, int c0, const f16* x1, int c1, const f16* gamma,
                                                   const f16* beta, f16* y, int S, int groups, int apply_silu, float eps) {
    __shared__ long long ls[64 * 2];   // fixed-point (sum, sum of squares) per group, see fx_add
    __shared__ unsigned mx[2];
    __shared__ float stat[64 * 2];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int C = c0 + c1, C8 = C / 8, cg = C / groups;
    if (tid < groups * 2) ls[tid] = 0;
    if (tid < 2) mx[tid] = 0u;
    __syncthreads();
    const int nchunk = S * C8;
    // every 8-channel chunk of the sample folded into its groups: emit(group, sum, sum of squares) per run of one group
    auto sweep = [&](auto&& emit) {
        for (int i = tid; i < nchunk; i += 256) {
            const int s = i / C8, cc = i - s * C8;
            U4H8 t;
            t.u = *reinterpret_cast<const uint4*>(gn_src(x0, c0, x1, c1, (size_t)b * S + s, cc * 8));
            int gcur = (cc * 8) / cg;
            float as = 0.f, aq = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int gi = (cc * 8 + k) / cg;
                if (gi != gcur) {
                    emit(gcur, as, aq);
                    as = aq = 0.f;
                    gcur = gi;
                }
                const float v = (float)t.e[k] - gn_shift(x0, c0, x1, c1, (size_t)b * S, gi, cg);
                as += v;
                aq += v * v;
            }
            emit(gcur, as, aq);
        }
    };
    {
        float ma = 0.f, mq = 0.f;
        sweep([&](int, float as, float aq) { ma = fmaxf(ma, fabsf(as)); mq = fmaxf(mq, aq); });
        atomicMax(&mx[0], __float_as_uint(ma));
        atomicMax(&mx[1], __float_as_uint(mq));
    }
    __syncthreads();
    const int ses = fx_scale_exp(mx[0]), seq = fx_scale_exp(mx[1]);
    sweep([&](int g, float as, float aq) { fx_add(&ls[g * 2], as, ses); fx_add(&ls[g * 2 + 1], aq, seq); });   // (the sample is L2-resident)
    __syncthreads();
    if (tid < groups) {
        const float inv = 1.0f / ((float)S * (float)cg);
        const float ms = fx_get(ls[tid * 2], ses) * inv;
        float var = fx_get(ls[tid * 2 + 1], seq) * inv - ms * ms;
        if (var < 0.f) var = 0.f;
        stat[tid * 2] = ms + gn_shift(x0, c0, x1, c1, (size_t)b * S, tid, cg);
        stat[tid * 2 + 1] = rsqrtf(var + eps);
    }
    __syncthreads();
    for (int i = tid; i < nchunk; i += 256) {
        const int s = i / C8, cc = i - s * C8;
        U4H8 t, ga, be, o;
        t.u = *reinterpret_cast<const uint4*>(gn_src(x0, c0, x1, c1, (size_t)b * S + s, cc * 8));
        ga.u = *reinterpret_cast<const uint4*>(gamma + (size_t)s * C + cc * 8);
        be.u = *reinterpret_cast<const uint4*>(beta + (size_t)s * C + cc * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int gi = (cc * 8 + k) / cg;
            float v = ((float)t.e[k] - stat[gi * 2]) * stat[gi * 2 + 1] * (float)ga.e[k] + (float)be.e[k];
            if (apply_silu) v = vd_silu(v);
            o.e[k] = (f16)v;
        }
        *reinterpret_cast<uint4*>(y + ((size_t)b * S + s) * C + cc * 8) = o.u;
    }
}

// LayerNorm: one wave per row, row kept in registers (two-pass variance), C <= 2048, C % 8 == 0
constexpr int LN_MAX_CH = 4;
__global__ __launch_bounds__(256) void layernorm_kernel(const f16* x, const f16* gamma, const f16* beta, f16* y,
                                                        int rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int C8 = C >> 3;
    const f16* xr = x + (size_t)row * C;
    float v[LN_MAX_CH][8];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAX_CH; ++j) {
        const int cc = lane + 64 * j;
        if (cc < C8) {
            U4H8 t;
            t.u = *reinterpret_cast<const uint4*>(xr + cc * 8);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v[j][i] = (float)t.e[i];
                s += v[j][i];
            }
        }
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAX_CH; ++j) {
        const int cc = lane + 64 * j;
        if (cc < C8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float dlt = v[j][i] - mean;
                q += dlt * dlt;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
    f16* yr = y + (size_t)row * C;
#pragma unroll
    for (int j = 0; j < LN_MAX_CH; ++j) {
        const int cc = lane + 64 * j;
        if (cc < C8) {
            U4H8 ga, be, o;
            ga.u = *reinterpret_cast<const uint4*>(gamma + cc * 8);
            be.u = *reinterpret_cast<const uint4*>(beta + cc * 8);
#pragma unroll
            for (int i = 0; i < 8; ++i) o.e[i] = (f16)((v[j][i] - mean) * rstd * (float)ga.e[i] + (float)be.e[i]);
            *reinterpret_cast<uint4*>(yr + cc * 8) = o.u;
        }
    }
}

}  // namespace

extern "C" size_t vd_groupnorm_workspace_bytes(int B, int HW, int C, int groups) {
    if (B <= 0 || HW <= 0 || C <= 0 || groups <= 0) return 0;
    const GnGeom g = gn_geom(HW, C);
    return ((size_t)B * g.nchunk * groups * 2 + (size_t)B * groups * 2) * sizeof(float);
}

extern "C" int vd_groupnorm_silu_f16(const void* x0, int c0, const void* x1, int c1, const void* gamma,
                                     const void* beta, void* y, float* stats, int B, int HW, int groups, float eps,
                                     int apply_silu, hipStream_t stream) {
    if (x1 == nullptr) c1 = 0;
    const int C = c0 + c1;
    VD_REQUIRE(x0 && gamma && beta && y && stats, "vd_groupnorm_silu_f16: null pointer");
    VD_REQUIRE(B > 0 && HW > 0 && C > 0, "vd_groupnorm_silu_f16: empty input");
    VD_REQUIRE(groups > 0 && groups <= 64 && C % groups == 0, "vd_groupnorm_silu_f16: bad groups=%d for C=%d", groups, C);
    VD_REQUIRE(c0 % 8 == 0 && c1 % 8 == 0, "vd_groupnorm_silu_f16: channel counts must be multiples of 8 (c0=%d c1=%d)", c0, c1);
    VD_REQUIRE(C <= GN_MAX_POS * 256 * 8, "vd_groupnorm_silu_f16: C=%d too large", C);
    const GnGeom g = gn_geom(HW, C);
    {   // low-resolution levels: one block per (sample, channel slab), statistics stay in the block
        const int cg = C / groups;
        int slab = cg;  // lcm(cg, 8)
        while (slab % 8 != 0) slab += cg;
        const int slab_chunks = slab / 8;
        if (C % slab == 0 && slab / cg <= 16 && slab_chunks <= 64 &&
            (HW + 256 / slab_chunks - 1) / (256 / slab_chunks) <= GN_SLAB_ITEMS) {
            const int items = (HW + 256 / slab_chunks - 1) / (256 / slab_chunks);  // rows per thread
#define VD_GN_SLAB(N_)                                                                                                     \
    hipLaunchKernelGGL(gn_slab_kernel<N_>, dim3(C / slab, B), dim3(256), 0, stream, (const f16*)x0, c0, (const f16*)x1, c1, \
                       (const f16*)gamma, (const f16*)beta, (f16*)y, HW, C, cg, slab_chunks, apply_silu,                    \
                       1.0f / ((float)HW * (float)cg), eps)
            if (items <= 2) VD_GN_SLAB(2);
            else if (items <= 6) VD_GN_SLAB(6);
            else VD_GN_SLAB(12);
#undef VD_GN_SLAB
            return vd_check_launch("vd_groupnorm_silu_f16");
        }
    }
    float* part = stats;
    hipLaunchKernelGGL(gn_partial_kernel, dim3(g.nchunk, B), dim3(256), 0, stream, (const f16*)x0, c0, (const f16*)x1,
                       c1, part, HW, groups, g);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(g.nchunk, B), dim3(256), 0, stream, (const f16*)x0, c0, (const f16*)x1, c1,
                       (const f16*)gamma, (const f16*)beta, part, (f16*)y, HW, groups, apply_silu,
                       1.0f / ((float)HW * (float)(C / groups)), eps, g);
    return vd_check_launch("vd_groupnorm_silu_f16");
}

namespace {
// GroupNorm as a per-(sample, channel) affine map: scale = rstd * gamma, shift = beta - mean * scale (fp16), from the slab
// partial sums of gn_partial_kernel -- for consumers that apply the normalisation themselves (vd_gemm_row320_chain_f16).
__global__ __launch_bounds__(256) void gn_affine_kernel(const f16* x0, const f16* gamma, const f16* beta, const float* part,
                                                        f16* sc, f16* sh, int HW, int C, int groups, int nchunk,
                                                        float inv_count, float eps) {
    const int tid = threadIdx.x, b = blockIdx.x;
    __shared__ float red[256];
    __shared__ float stat[64 * 2];
    const int ne = groups * 2;
    const int e = tid % ne, lanes = 256 / ne, cl = tid / ne;
    float acc = 0.f;
    if (cl < lanes) {
        const float* pp = part + (size_t)b * nchunk * ne + e;
        int c = cl;
        for (; c + 7 * lanes < nchunk; c += 8 * lanes) {  // 8 independent loads in flight: this is a latency chain
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = pp[(size_t)(c + u * lanes) * ne];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += t[u];
        }
        for (; c < nchunk; c += lanes) acc += pp[(size_t)c * ne];
    }
    red[tid] = (cl < lanes) ? acc : 0.f;
    __syncthreads();
    if (tid < ne) {
        float t = 0.f;
        for (int l = 0; l < lanes; ++l) t += red[l * ne + tid];
        red[tid] = t;
    }
    __syncthreads();
    if (tid < groups) {
        const float ms = red[tid * 2] * inv_count;   // mean of (x - k)
        float var = red[tid * 2 + 1] * inv_count - ms * ms;
        if (var < 0.f) var = 0.f;
        stat[tid * 2] = ms + (float)x0[(size_t)b * HW * C + tid * (C / groups)];   // the group's shift k_g (gn_shift)
        stat[tid * 2 + 1] = rsqrtf(var + eps);
    }
    __syncthreads();
    const int cg = C / groups;
    for (int c = tid; c < C; c += 256) {
        const float mean = stat[(c / cg) * 2], rstd = stat[(c / cg) * 2 + 1];
        const float s = rstd * (float)gamma[c];
        sc[(size_t)b * C + c] = (f16)s;
        sh[(size_t)b * C + c] = (f16)((float)beta[c] - mean * s);
    }
}
}  // namespace

extern "C" int vd_groupnorm_affine_f16(const void* x, const void* gamma, const void* beta, void* scale, void* shift,
                                       float* stats, int B, int HW, int C, int groups, float eps, hipStream_t stream) {
    VD_REQUIRE(x && gamma && beta && scale && shift && stats, "vd_groupnorm_affine_f16: null pointer");
    VD_REQUIRE(B > 0 && HW > 0 && C > 0, "vd_groupnorm_affine_f16: empty input");
    VD_REQUIRE(groups > 0 && groups <= 64 && C % groups == 0 && C % 8 == 0 && C <= GN_MAX_POS * 256 * 8,
               "vd_groupnorm_affine_f16: bad groups=%d / C=%d", groups, C);
    const GnGeom g = gn_geom(HW, C);
    hipLaunchKernelGGL(gn_partial_kernel, dim3(g.nchunk, B), dim3(256), 0, stream, (const f16*)x, C, (const f16*)nullptr, 0,
                       stats, HW, groups, g);
    hipLaunchKernelGGL(gn_affine_kernel, dim3(B), dim3(256), 0, stream, (const f16*)x, (const f16*)gamma, (const f16*)beta,
                       (const float*)stats, (f16*)scale, (f16*)shift, HW, C, groups, g.nchunk,
                       1.0f / ((float)HW * (float)(C / groups)), eps);
    return vd_check_launch("vd_groupnorm_affine_f16");
}

extern "C" int vd_groupnorm0d_silu_f16(const void* x0, int c0, const void* x1, int c1, const void* gamma, const void* beta,
                                       void* y, int B, int S, int groups, float eps, int apply_silu, hipStream_t stream) {
    if (x1 == nullptr) c1 = 0;
    const int C = c0 + c1;
    VD_REQUIRE(x0 && gamma && beta && y, "vd_groupnorm0d_silu_f16: null pointer");
    VD_REQUIRE(B > 0 && S > 0 && C > 0, "vd_groupnorm0d_silu_f16: empty input");
    VD_REQUIRE(groups > 0 && groups <= 64 && C % groups == 0, "vd_groupnorm0d_silu_f16: bad groups=%d for C=%d", groups, C);
    VD_REQUIRE(c0 % 8 == 0 && c1 % 8 == 0, "vd_groupnorm0d_silu_f16: channel counts must be multiples of 8 (c0=%d c1=%d)", c0, c1);
    hipLaunchKernelGGL(gn0d_kernel, dim3(B), dim3(256), 0, stream, (const f16*)x0, c0, (const f16*)x1, c1, (const f16*)gamma,
                       (const f16*)beta, (f16*)y, S, groups, apply_silu, eps);
    return vd_check_launch("vd_groupnorm0d_silu_f16");
}

namespace {
// Row statistics for the LayerNorm fold of vd_gemm_f16: 16 lanes per row (4 rows per wave, 16 per block), every 16-byte
// load of a row issued before the first is consumed, mean first, then the centred sum of squares from registers.
template <int NCH>
__global__ __launch_bounds__(256) void row_stats_kernel(const f16* x, float* stats, long rows, int C, long ldx, float eps) {
    const int sub = threadIdx.x & 15;
    const long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const long r = row < rows ? row : rows - 1;
    const int C8 = C >> 3;
    const f16* xr = x + r * ldx;
    U4H8 t[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int cc = sub + 16 * j;
        t[j].u = cc < C8 ? *reinterpret_cast<const uint4*>(xr + cc * 8) : make_uint4(0, 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) s += (float)t[j].e[i];      // lanes past C hold zeros
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j)
        if (sub + 16 * j < C8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float dlt = (float)t[j].e[i] - mean;
                q += dlt * dlt;
            }
        }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) q += __shfl_xor(q, o, 64);
    if (sub == 0 && row < rows) reinterpret_cast<float2*>(stats)[row] = make_float2(mean, rsqrtf(q / (float)C + eps));
}
}  // namespace

extern "C" int vd_row_stats_f16(const void* x, float* stats, int64_t rows, int C, int64_t ldx, float eps, hipStream_t stream) {
    VD_REQUIRE(x && stats && rows > 0 && C > 0 && C % 8 == 0 && C <= 2048 && ldx >= C && ldx % 8 == 0,
               "vd_row_stats_f16: bad arguments (rows=%ld C=%d ldx=%ld; C %% 8 == 0, C <= 2048)", (long)rows, C, (long)ldx);
    const dim3 grid((unsigned)((rows + 15) / 16)), block(256);
    const int nch = (C / 8 + 15) / 16;
    if (nch <= 3) hipLaunchKernelGGL(row_stats_kernel<3>, grid, block, 0, stream, (const f16*)x, stats, (long)rows, C, (long)ldx, eps);
    else if (nch <= 5) hipLaunchKernelGGL(row_stats_kernel<5>, grid, block, 0, stream, (const f16*)x, stats, (long)rows, C, (long)ldx, eps);
    else if (nch <= 10) hipLaunchKernelGGL(row_stats_kernel<10>, grid, block, 0, stream, (const f16*)x, stats, (long)rows, C, (long)ldx, eps);
    else hipLaunchKernelGGL(row_stats_kernel<16>, grid, block, 0, stream, (const f16*)x, stats, (long)rows, C, (long)ldx, eps);
    return vd_check_launch("vd_row_stats_f16");
}

extern "C" int vd_layernorm_f16(const void* x, const void* gamma, const void* beta, void* y, int rows, int C,
                                float eps, hipStream_t stream) {
    VD_REQUIRE(x && gamma && beta && y, "vd_layernorm_f16: null pointer");
    VD_REQUIRE(rows > 0 && C > 0, "vd_layernorm_f16: empty input");
    VD_REQUIRE(C % 8 == 0 && C <= LN_MAX_CH * 64 * 8, "vd_layernorm_f16: C=%d unsupported (need C%%8==0, C<=2048)", C);
    hipLaunchKernelGGL(layernorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, (const f16*)x, (const f16*)gamma,
                       (const f16*)beta, (f16*)y, rows, C, eps);
    return vd_check_launch("vd_layernorm_f16");
}
