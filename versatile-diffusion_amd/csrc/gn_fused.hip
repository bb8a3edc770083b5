// GroupNorm without its own passes over memory (round 4).
//
// The reference runs GroupNorm32 -> SiLU -> conv as three ops per ResBlock half (lib/model_zoo/openaimodel.py:196-200,
// 230-237,254-274; normalization(): diffusion_utils.py:175-191; Normalize in front of SpatialTransformer.proj_in:
// attention.py:76-77,255-259).  Rounds 1-3 ran it as one HBM-bound kernel pair per norm (statistics pass + apply pass,
// csrc/norm.hip).  Here the norm is split along its data dependences instead:
//   1. STATISTICS come from whoever PRODUCES the tensor: the epilogues of conv3x3_halo_kernel / gemm_f16_kernel and the
//      split-K reduce emit, per output channel and per block of `R` rows of one image, the pair (mean, M2 = sum (x - mean)^2)
//      of the values they store (VdGemmDesc.out_stats).  Per CHANNEL, so a consumer can fold the channels of one tensor --
//      or of the two tensors of a skip concat -- into its 32 groups whatever the producers' tile shapes were; a tensor that
//      is consumed twice (every skip connection) is measured once.  vd_chan_stats_f16 computes the same partials with one
//      read of x for tensors whose producer does not (test reference and fallback).
//   2. vd_gn_table_f32 folds the partials of one (sample, group) with Chan's parallel-variance update and writes the
//      normalisation as a per-(sample, channel) affine map  y = x * scale + shift,  scale = rstd * gamma,
//      shift = beta - mean * rstd * gamma  (fp32 [B][2][C]): a few KB, one tiny launch.
//   3. The map (+ SiLU) is applied by the CONSUMER: conv3x3_halo_kernel transforms the input halo in LDS once per
//      64-channel chunk (VdGemmDesc.in_norm, conv_halo_kernel.h), so the normalised activation never exists in HBM;
//      vd_gn_apply_table_f16 is the one-read one-write elementwise form for consumers without that path.
// Numerics: partial sums run over (x - k) with k = the block's first row of that channel (a sample, hence within a few
// sigma of the mean), so M2 = S2 - S1^2 / n does not cancel when |mean| >> sigma; partials are combined as
// M2 = sum M2_i + sum n_i (mean_i - mean)^2 in fp32 (torch's Welford / two-pass robustness, see
// tests/test_kernels_gpu.py::test_gn_fused_large_mean_small_spread).
#include <stdlib.h>
#include "vd_common.h"
#include "../../include/vd_hip.h"

namespace {

// ---- 1. per-channel partial statistics of x [M][C] (row stride ldx): one partial per R consecutive rows ---------------
// grid (ceil(C / 64), M / R), 256 threads = 8 channel octets x 32 row lanes
__global__ __launch_bounds__(256) void chan_stats_kernel(const f16* __restrict__ x, int ldx, int C, int R, float2* __restrict__ out) {
    __shared__ float red[32][64][2];
    const int tid = threadIdx.x, co = tid & 7, rl = tid >> 3;
    const int c = blockIdx.x * 64 + co * 8;
    const size_t row0 = (size_t)blockIdx.y * R;
    float s[8], q[8], k[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = q[i] = k[i] = 0.f;
    if (c < C) {
        U4H8 p;
        p.u = *reinterpret_cast<const uint4*>(x + row0 * ldx + c);
#pragma unroll
        for (int i = 0; i < 8; ++i) k[i] = (float)p.e[i];
        for (int r = rl; r < R; r += 128) {   // 4 loads in flight
            U4H8 t[4];
            float w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ru = r + 32 * u;
                w[u] = ru < R ? 1.f : 0.f;
                t[u].u = *reinterpret_cast<const uint4*>(x + (row0 + (ru < R ? ru : r)) * ldx + c);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float v = ((float)t[u].e[i] - k[i]) * w[u];
                    s[i] += v;
                    q[i] += v * v;
                }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        red[rl][co * 8 + i][0] = s[i];
        red[rl][co * 8 + i][1] = q[i];
    }
    __syncthreads();
    if (tid < 64) {
        const int ch = blockIdx.x * 64 + tid;
        if (ch < C) {
            float S = 0.f, Q = 0.f;
#pragma unroll 8
            for (int l = 0; l < 32; ++l) {
                S += red[l][tid][0];
                Q += red[l][tid][1];
            }
            const float kk = (float)x[row0 * ldx + ch];
            const float n = (float)R;
            out[(size_t)blockIdx.y * C + ch] = make_float2(kk + S / n, fmaxf(Q - S * S / n, 0.f));
        }
    }
}

// ---- 2. partials -> per-(sample, channel) affine map --------------------------------------------------------------------
struct GnTableArgs {
    const float2* st0; int T0, c0;   // source 0: [B * T0][c0] partials of HW / T0 rows each
    const float2* st1; int T1, c1;   // optional source 1 (channel concat)
    const f16* gamma; const f16* beta;
    int HW, groups;
    float eps;
    float* table;                    // [B][2][c0 + c1] (or NULL)
    f16* scale16; f16* shift16;      // optional fp16 [B][c0 + c1] copies (operands of vd_gemm_row320_chain_f16)
    f16* center16;                   // optional, with scale16: fp16(mean) per channel; shift16 is then relative to it (see below)
};

// grid (groups, B), ONE wave per (sample, group): the cg x T partials of the group are folded in a single pass around a pivot
// (the group's first partial mean: sum n_i (mean_i - p), sum M2_i + n_i (mean_i - p)^2 -- no second pass, no LDS, no barrier;
// the spread of channel means inside a group is part of the group's variance, so the pivot form does not cancel)
__global__ __launch_bounds__(64) void gn_table_kernel(const GnTableArgs a) {
    const int g = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int C = a.c0 + a.c1, cg = C / a.groups;
    const int Tmax = a.T0 > a.T1 ? a.T0 : a.T1;
    const float n0 = (float)(a.HW / a.T0), n1 = a.T1 > 0 ? (float)(a.HW / a.T1) : 0.f;
    const int chp = g * cg;
    const int items = cg * Tmax;
    // ONE round trip to memory: every lane requests its partials (and its gamma / beta) before anything is consumed; the pivot
    // is lane 0's first partial (item 0 = first channel of the group, block 0), broadcast from its register
    constexpr int KI = 8;   // items per lane held in registers (cg * T <= 512: every shape of the UNet / VAE); deeper lists loop
    float2 pv[KI];
    float pn[KI];
#pragma unroll
    for (int k = 0; k < KI; ++k) {
        const int idx = lane + k * 64;
        pn[k] = 0.f;
        pv[k] = make_float2(0.f, 0.f);
        if (idx < items) {
            const int cl = idx / Tmax, t = idx - cl * Tmax;
            const int ch = chp + cl;
            if (ch < a.c0) {
                if (t < a.T0) { pn[k] = n0; pv[k] = a.st0[((size_t)b * a.T0 + t) * a.c0 + ch]; }
            } else {
                if (t < a.T1) { pn[k] = n1; pv[k] = a.st1[((size_t)b * a.T1 + t) * a.c1 + (ch - a.c0)]; }
            }
        }
    }
    float gam[2] = {0.f, 0.f}, bet[2] = {0.f, 0.f};   // cg <= 128 channels per group
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (lane + k * 64 < cg) {
            gam[k] = (float)a.gamma[chp + lane + k * 64];
            bet[k] = (float)a.beta[chp + lane + k * 64];
        }
    const float pivot = __shfl(pv[0].x, 0, 64);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < KI; ++k) {
        const float dm = pv[k].x - pivot;
        s1 += pn[k] * dm;
        s2 += pn[k] > 0.f ? pv[k].y + pn[k] * dm * dm : 0.f;
    }
    for (int idx = lane + KI * 64; idx < items; idx += 64) {
        const int cl = idx / Tmax, t = idx - cl * Tmax;
        const int ch = chp + cl;
        float n;
        float2 p;
        if (ch < a.c0) {
            if (t >= a.T0) continue;
            n = n0;
            p = a.st0[((size_t)b * a.T0 + t) * a.c0 + ch];
        } else {
            if (t >= a.T1) continue;
            n = n1;
            p = a.st1[((size_t)b * a.T1 + t) * a.c1 + (ch - a.c0)];
        }
        const float dm = p.x - pivot;
        s1 += n * dm;
        s2 += p.y + n * dm * dm;
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    const float ntot = (float)a.HW * (float)cg;
    const float dmean = s1 / ntot;
    const float mean = pivot + dmean;
    const float var = fmaxf(s2 / ntot - dmean * dmean, 0.f);
    const float rstd = rsqrtf(var + a.eps);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = lane + k * 64;
        if (i < cg) {
            const int ch = chp + i;
            const float sc = rstd * gam[k];
            const float sh = bet[k] - mean * sc;
            if (a.table) {
                a.table[((size_t)b * 2) * C + ch] = sc;
                a.table[((size_t)b * 2 + 1) * C + ch] = sh;
            }
            if (a.scale16) {
                // fp16 operands: x * scale + shift loses |mean| / sigma * 2^-11 to the rounding of scale and shift (both O(|mean| / sigma)).
                // Centred form: (x - c) * scale + shift' with c = fp16(mean) -- x - c is exact in fp16 near the mean, and
                // shift' = beta - (mean - c) * scale is O(1), so the error is 2^-11 of the NORMALISED value whatever the offset
                a.scale16[(size_t)b * C + ch] = (f16)sc;
                if (a.center16) {
                    const f16 c16 = (f16)mean;
                    a.center16[(size_t)b * C + ch] = c16;
                    a.shift16[(size_t)b * C + ch] = (f16)(bet[k] - (mean - (float)c16) * sc);
                } else {
                    a.shift16[(size_t)b * C + ch] = (f16)sh;
                }
            }
        }
    }
}

// ---- 3. elementwise consumer: out[b][p][c] = act(x[b][p][c] * scale[b][c] + shift[b][c]), x = cat(x0, x1) on channels ----
// A thread keeps ONE channel octet (two where C > 2048) for all its rows, so scale / shift are read once per block and the
// row loop is a 16-byte load, 8 FMAs (+ SiLU) and a 16-byte store; consecutive threads cover a whole row (full cache
// lines).  grid (row chunks, B), 256 threads = TC channel octets x R row lanes.
struct GnApplyArgs {
    const f16* x0; const f16* x1; const float* table; f16* out;
    int c0, c1, HW, TC, R, npos, rows_per_chunk, silu;
    // SUMS: the map is built here from the producers' per-(image, channel) fixed-point sums (VdGemmDesc.stat_sums, ABI 7)
    const long long* sums0; const long long* sums1; const f16* gamma; const f16* beta;
    int groups; float eps;
};

// SUMS = true: no table -- 8 lanes per group add the group's channel sums of this image (int64 -> fp64: exact integers, one
// rounding at the end), (mean, rstd) of the 32 groups meet in LDS, a thread then forms scale / shift of its octets in registers
template <bool SUMS>
__global__ __launch_bounds__(256) void gn_apply_table_kernel(const GnApplyArgs a) {
    __shared__ float gstat[32][2];
    const int tid = threadIdx.x, b = blockIdx.y;
    const int C = a.c0 + a.c1, C8 = C / 8;
    const int tc = tid % a.TC, rl = tid / a.TC;
    const int cg = SUMS ? C / a.groups : 1;
    const int r0 = blockIdx.x * a.rows_per_chunk;
    int r1 = r0 + a.rows_per_chunk;
    if (r1 > a.HW) r1 = a.HW;
    const size_t rowb = (size_t)b * a.HW;
    // SUMS: the first batch of rows, gamma and beta of the first octet are requested BEFORE the fold (they do not depend on it), so
    // the block pays max(fold, row latency), not their sum
    U4H8 pre[4], gm0, bt0;
    if constexpr (SUMS) {
        if (rl < a.R) {
            const int ch = tc * 8;
            const bool second = ch >= a.c0;
            const f16* src = second ? a.x1 + (ch - a.c0) : a.x0 + ch;
            const int ld = second ? a.c1 : a.c0;
            const int r = r0 + rl;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ru = r + u * a.R;
                pre[u].u = *reinterpret_cast<const uint4*>(src + (rowb + (ru < r1 ? ru : (r < r1 ? r : r0))) * ld);
            }
            gm0.u = *reinterpret_cast<const uint4*>(a.gamma + ch);
            bt0.u = *reinterpret_cast<const uint4*>(a.beta + ch);
        }
        const int g = tid >> 3, sub = tid & 7;
        double s = 0.0, q = 0.0;
        if (g < a.groups) {
#pragma unroll 4
            for (int i = sub; i < cg; i += 8) {
                const int ch = g * cg + i;
                const long long* p = ch < a.c0 ? a.sums0 + ((size_t)b * a.c0 + ch) * 2 : a.sums1 + ((size_t)b * a.c1 + (ch - a.c0)) * 2;
                const longlong2 v = *reinterpret_cast<const longlong2*>(p);
                s += (double)v.x;
                q += (double)v.y;
            }
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            s += __shfl_xor(s, o, 64);
            q += __shfl_xor(q, o, 64);
        }
        if (sub == 0 && g < a.groups) {
            const double n = (double)a.HW * (double)cg;
            const double mean = s / (4294967296.0 * n);
            const double var = fmax(q / (65536.0 * n) - mean * mean, 0.0);
            gstat[g][0] = (float)mean;
            gstat[g][1] = rsqrtf((float)var + a.eps);
        }
        __syncthreads();
    }
    if (rl >= a.R) return;
    for (int pos = 0; pos < a.npos; ++pos) {
        const int cc = tc + pos * a.TC;
        if (cc >= C8) break;
        const int ch = cc * 8;
        float sc[8], sf[8];
        if constexpr (SUMS) {
            U4H8 gm = gm0, bt = bt0;
            if (pos > 0) {
                gm.u = *reinterpret_cast<const uint4*>(a.gamma + ch);
                bt.u = *reinterpret_cast<const uint4*>(a.beta + ch);
            }
            int g = ch / cg, left = cg - (ch - g * cg);   // channels of group g from ch on
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (left == 0) { ++g; left = cg; }
                --left;
                sc[q] = gstat[g][1] * (float)gm.e[q];
                sf[q] = (float)bt.e[q] - gstat[g][0] * sc[q];
            }
        } else {
            const float* tb = a.table + ((size_t)b * 2) * C + ch;
            const float4 s0 = *reinterpret_cast<const float4*>(tb), s1 = *reinterpret_cast<const float4*>(tb + 4);
            const float4 h0 = *reinterpret_cast<const float4*>(tb + C), h1 = *reinterpret_cast<const float4*>(tb + C + 4);
            sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
            sf[0] = h0.x; sf[1] = h0.y; sf[2] = h0.z; sf[3] = h0.w; sf[4] = h1.x; sf[5] = h1.y; sf[6] = h1.z; sf[7] = h1.w;
        }
        const bool second = ch >= a.c0;
        const f16* src = second ? a.x1 + (ch - a.c0) : a.x0 + ch;
        const int ld = second ? a.c1 : a.c0;
        for (int r = r0 + rl; r < r1; r += 4 * a.R) {   // 4 loads in flight
            U4H8 t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ru = r + u * a.R;
                if (SUMS && pos == 0 && r == r0 + rl) t[u] = pre[u];
                else t[u].u = *reinterpret_cast<const uint4*>(src + (rowb + (ru < r1 ? ru : r)) * ld);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ru = r + u * a.R;
                if (ru < r1) {
                    U4H8 o;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float y = fmaf((float)t[u].e[q], sc[q], sf[q]);
                        o.e[q] = (f16)(a.silu ? y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896f * y)) : y);
                    }
                    *reinterpret_cast<uint4*>(a.out + (rowb + ru) * C + ch) = o.u;
                }
            }
        }
    }
}

// ---- 2 + 3 in one launch: partials -> (mean, rstd) of the block's groups -> normalise (+SiLU) a [rows x slab] panel ------
// A block owns one sample, a SLAB of S = lcm(channels per group, 8) channels (whole groups, whole 16-byte octets) and a range
// of rows; it first folds the partials of its S / cg groups (T x S pairs, L2-resident), then streams its panel once.
struct GnFromStatsArgs {
    const f16* x0; const float2* st0; int T0, c0;
    const f16* x1; const float2* st1; int T1, c1;
    const f16* gamma; const f16* beta; f16* out;
    int HW, groups, S, rows_per_block, silu;
    float eps;
};

constexpr int GFS_MAXG = 8;   // groups per slab (S / cg): 4 at cg = 10, 2 at cg = 20 / 60, 1 at cg = 40 / 80, 4 at cg = 30

// sum of v[0 .. GFS_MAXG) over the 256 threads of the block, in a fixed order (no atomics: run-to-run identical)
__device__ __forceinline__ void block_sum_groups(float* v, float (*sh)[GFS_MAXG], float* out, int ng) {
#pragma unroll
    for (int g = 0; g < GFS_MAXG; ++g)
        if (g < ng) v[g] = wave_sum(v[g]);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int g = 0; g < GFS_MAXG; ++g) sh[threadIdx.x >> 6][g] = v[g];
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < GFS_MAXG; ++g) out[g] = sh[0][g] + sh[1][g] + sh[2][g] + sh[3][g];
}

// grid (row ranges, slabs, B), 256 threads = S / 8 channel octets x row lanes
__global__ __launch_bounds__(256) void gn_from_stats_kernel(const GnFromStatsArgs a) {
    __shared__ float sh[4][GFS_MAXG];
    const int tid = threadIdx.x, b = blockIdx.z, slab = blockIdx.y;
    const int C = a.c0 + a.c1, cg = C / a.groups, ng = a.S / cg;
    const int ch0 = slab * a.S;
    const int Tmax = a.T0 > a.T1 ? a.T0 : a.T1;
    const float n0 = (float)(a.HW / a.T0), n1 = a.T1 > 0 ? (float)(a.HW / a.T1) : 0.f;
    auto partial = [&](int idx, int& gl, float& n, float2& p) {
        const int cl = idx / Tmax, t = idx - cl * Tmax;
        const int ch = ch0 + cl;
        gl = cl / cg;
        if (ch < a.c0) {
            if (t >= a.T0) return false;
            n = n0;
            p = a.st0[((size_t)b * a.T0 + t) * a.c0 + ch];
        } else {
            if (t >= a.T1) return false;
            n = n1;
            p = a.st1[((size_t)b * a.T1 + t) * a.c1 + (ch - a.c0)];
        }
        return true;
    };
    auto add_to = [&](float* v, int gl, float x) {   // v[gl] += x with a compile-time register index
#pragma unroll
        for (int g = 0; g < GFS_MAXG; ++g)
            if (g == gl) v[g] += x;
    };
    const int items = a.S * Tmax;
    const float ntot = (float)a.HW * (float)cg;
    float acc[GFS_MAXG], mean[GFS_MAXG], m2[GFS_MAXG];
#pragma unroll
    for (int g = 0; g < GFS_MAXG; ++g) acc[g] = 0.f;
    for (int idx = tid; idx < items; idx += 256) {
        int gl; float n; float2 p;
        if (partial(idx, gl, n, p)) add_to(acc, gl, n * p.x);
    }
    block_sum_groups(acc, sh, mean, ng);
#pragma unroll
    for (int g = 0; g < GFS_MAXG; ++g) {
        mean[g] /= ntot;
        acc[g] = 0.f;
    }
    for (int idx = tid; idx < items; idx += 256) {   // second pass over the (L2-resident) partials
        int gl; float n; float2 p;
        if (partial(idx, gl, n, p)) {
            float mg = 0.f;
#pragma unroll
            for (int g = 0; g < GFS_MAXG; ++g)
                if (g == gl) mg = mean[g];
            const float dm = p.x - mg;
            add_to(acc, gl, p.y + n * dm * dm);
        }
    }
    block_sum_groups(acc, sh, m2, ng);
    const int TC = a.S / 8, RL = 256 / TC;
    const int tc = tid % TC, rl = tid / TC;
    if (rl >= RL) return;
    const int ch = ch0 + tc * 8;
    float sc[8], sf[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int gl = (tc * 8 + q) / cg;
        float mg = 0.f, vg = 0.f;
#pragma unroll
        for (int g = 0; g < GFS_MAXG; ++g)
            if (g == gl) { mg = mean[g]; vg = m2[g]; }
        const float rstd = rsqrtf(vg / ntot + a.eps);
        sc[q] = rstd * (float)a.gamma[ch + q];
        sf[q] = (float)a.beta[ch + q] - mg * sc[q];
    }
    const bool second = ch >= a.c0;
    const f16* src = second ? a.x1 + (ch - a.c0) : a.x0 + ch;
    const int ld = second ? a.c1 : a.c0;
    const int r0 = blockIdx.x * a.rows_per_block;
    int r1 = r0 + a.rows_per_block;
    if (r1 > a.HW) r1 = a.HW;
    const size_t rowb = (size_t)b * a.HW;
    for (int r = r0 + rl; r < r1; r += 4 * RL) {   // 4 loads in flight
        U4H8 t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ru = r + u * RL;
            t[u].u = *reinterpret_cast<const uint4*>(src + (rowb + (ru < r1 ? ru : r)) * ld);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ru = r + u * RL;
            if (ru < r1) {
                U4H8 o;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float y = fmaf((float)t[u].e[q], sc[q], sf[q]);
                    o.e[q] = (f16)(a.silu ? y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896f * y)) : y);
                }
                *reinterpret_cast<uint4*>(a.out + (rowb + ru) * C + ch) = o.u;
            }
        }
    }
}

}  // namespace

extern "C" int vd_groupnorm_from_stats_f16(const void* x0, int c0, const float* stats0, int T0, const void* x1, int c1,
                                           const float* stats1, int T1, const void* gamma, const void* beta, void* y, int B, int HW,
                                           int groups, float eps, int apply_silu, hipStream_t stream) {
    VD_REQUIRE(x0 && stats0 && gamma && beta && y, "vd_groupnorm_from_stats_f16: null pointer");
    if (!x1) { c1 = 0; T1 = 0; stats1 = nullptr; }
    VD_REQUIRE(x1 == nullptr || stats1 != nullptr, "vd_groupnorm_from_stats_f16: x1 without stats1");
    VD_REQUIRE(B > 0 && B <= 65535 && HW > 0 && groups > 0 && c0 > 0 && c0 % 8 == 0 && c1 % 8 == 0, "vd_groupnorm_from_stats_f16: bad sizes");
    const int C = c0 + c1;
    VD_REQUIRE(C % groups == 0, "vd_groupnorm_from_stats_f16: C=%d must divide into %d groups", C, groups);
    VD_REQUIRE(T0 > 0 && HW % T0 == 0 && (c1 == 0 || (T1 > 0 && HW % T1 == 0)), "vd_groupnorm_from_stats_f16: partials must tile the %d rows of a sample (T0=%d T1=%d)", HW, T0, T1);
    const int cg = C / groups;
    int S = cg;   // lcm(cg, 8)
    while (S % 8 != 0) S += cg;
    VD_REQUIRE(C % S == 0 && S / cg <= GFS_MAXG && S / 8 <= 256, "vd_groupnorm_from_stats_f16: C=%d with %d groups has no slab (S=%d)", C, groups, S);
    GnFromStatsArgs a;
    a.x0 = reinterpret_cast<const f16*>(x0); a.st0 = reinterpret_cast<const float2*>(stats0); a.T0 = T0; a.c0 = c0;
    a.x1 = reinterpret_cast<const f16*>(x1); a.st1 = reinterpret_cast<const float2*>(stats1); a.T1 = T1; a.c1 = c1;
    a.gamma = reinterpret_cast<const f16*>(gamma); a.beta = reinterpret_cast<const f16*>(beta); a.out = reinterpret_cast<f16*>(y);
    a.HW = HW; a.groups = groups; a.S = S; a.silu = apply_silu; a.eps = eps;
    // row ranges: enough blocks to cover the chip a few times, panels of >= 64 rows so the partial fold stays a small share
    const int slabs = C / S;
    int ranges = (1536 + B * slabs - 1) / (B * slabs);
    int rpb = (HW + ranges - 1) / ranges;
    if (rpb < 64) rpb = 64;
    if (rpb > HW) rpb = HW;
    a.rows_per_block = rpb;
    ranges = (HW + rpb - 1) / rpb;
    hipLaunchKernelGGL(gn_from_stats_kernel, dim3(ranges, slabs, B), dim3(256), 0, stream, a);
    return vd_check_launch("vd_groupnorm_from_stats_f16");
}

extern "C" int vd_chan_stats_f16(const void* x, long M, int C, int ldx, int rows_per_partial, float* stats, hipStream_t stream) {
    VD_REQUIRE(x && stats, "vd_chan_stats_f16: null pointer");
    VD_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && ldx >= C && ldx % 8 == 0, "vd_chan_stats_f16: M=%ld C=%d ldx=%d (C, ldx multiples of 8)", M, C, ldx);
    VD_REQUIRE(rows_per_partial > 0 && M % rows_per_partial == 0, "vd_chan_stats_f16: M=%ld is not a multiple of rows_per_partial=%d", M, rows_per_partial);
    VD_REQUIRE(M / rows_per_partial <= 65535, "vd_chan_stats_f16: %ld partials > 65535", M / rows_per_partial);
    hipLaunchKernelGGL(chan_stats_kernel, dim3((C + 63) / 64, (unsigned)(M / rows_per_partial)), dim3(256), 0, stream,
                       reinterpret_cast<const f16*>(x), ldx, C, rows_per_partial, reinterpret_cast<float2*>(stats));
    return vd_check_launch("vd_chan_stats_f16");
}

static int gn_table_launch(const float* stats0, int T0, int c0, const float* stats1, int T1, int c1, int B, int HW, const void* gamma,
                           const void* beta, int groups, float eps, float* table, void* scale16, void* shift16, void* center16, hipStream_t stream);

extern "C" int vd_gn_table_f32(const float* stats0, int T0, int c0, const float* stats1, int T1, int c1, int B, int HW,
                               const void* gamma, const void* beta, int groups, float eps, float* table, hipStream_t stream) {
    VD_REQUIRE(table, "vd_gn_table_f32: null pointer");
    return gn_table_launch(stats0, T0, c0, stats1, T1, c1, B, HW, gamma, beta, groups, eps, table, nullptr, nullptr, nullptr, stream);
}

extern "C" int vd_gn_affine_from_stats_f16(const float* stats0, int T0, int c0, const float* stats1, int T1, int c1, int B, int HW,
                                           const void* gamma, const void* beta, int groups, float eps, void* scale, void* shift,
                                           void* center, hipStream_t stream) {
    VD_REQUIRE(scale && shift, "vd_gn_affine_from_stats_f16: null pointer");
    return gn_table_launch(stats0, T0, c0, stats1, T1, c1, B, HW, gamma, beta, groups, eps, nullptr, scale, shift, center, stream);
}

static int gn_table_launch(const float* stats0, int T0, int c0, const float* stats1, int T1, int c1, int B, int HW, const void* gamma,
                           const void* beta, int groups, float eps, float* table, void* scale16, void* shift16, void* center16, hipStream_t stream) {
    VD_REQUIRE(stats0 && gamma && beta, "vd_gn_table_f32: null pointer");
    if (!stats1) { T1 = 0; c1 = 0; }
    VD_REQUIRE(B > 0 && B <= 65535 && HW > 0 && groups > 0 && c0 > 0 && c1 >= 0, "vd_gn_table_f32: bad sizes");
    VD_REQUIRE((c0 + c1) % groups == 0 && (c0 + c1) % 4 == 0 && (c0 + c1) / groups <= 128, "vd_gn_table_f32: C=%d must divide into %d groups of at most 128 channels", c0 + c1, groups);
    VD_REQUIRE(T0 > 0 && HW % T0 == 0 && (c1 == 0 || (T1 > 0 && HW % T1 == 0)), "vd_gn_table_f32: partials must tile the %d rows of a sample (T0=%d T1=%d)", HW, T0, T1);
    GnTableArgs a;
    a.st0 = reinterpret_cast<const float2*>(stats0); a.T0 = T0; a.c0 = c0;
    a.st1 = reinterpret_cast<const float2*>(stats1); a.T1 = T1; a.c1 = c1;
    a.gamma = reinterpret_cast<const f16*>(gamma); a.beta = reinterpret_cast<const f16*>(beta);
    a.HW = HW; a.groups = groups; a.eps = eps; a.table = table;
    a.scale16 = reinterpret_cast<f16*>(scale16); a.shift16 = reinterpret_cast<f16*>(shift16); a.center16 = reinterpret_cast<f16*>(center16);
    hipLaunchKernelGGL(gn_table_kernel, dim3(groups, B), dim3(64), 0, stream, a);
    return vd_check_launch("vd_gn_table_f32");
}

static int gn_apply_launch(const char* who, GnApplyArgs& a, const void* x0, int c0, const void* x1, int c1, int B, int HW, int silu, void* out,
                           hipStream_t stream);

extern "C" int vd_gn_apply_table_f16(const void* x0, int c0, const void* x1, int c1, int B, int HW, const float* table, int silu,
                                     void* out, hipStream_t stream) {
    VD_REQUIRE(x0 && table && out, "vd_gn_apply_table_f16: null pointer");
    GnApplyArgs a;
    a.table = table;
    a.sums0 = a.sums1 = nullptr; a.gamma = a.beta = nullptr; a.groups = 0; a.eps = 0.f;
    return gn_apply_launch("vd_gn_apply_table_f16", a, x0, c0, x1, c1, B, HW, silu, out, stream);
}

extern "C" int vd_gn_apply_sums_f16(const void* x0, int c0, const void* sums0, const void* x1, int c1, const void* sums1, int B, int HW,
                                    const void* gamma, const void* beta, int groups, float eps, int silu, void* out, hipStream_t stream) {
    VD_REQUIRE(x0 && sums0 && gamma && beta && out && (!x1 || sums1), "vd_gn_apply_sums_f16: null pointer");
    if (!x1) c1 = 0;
    VD_REQUIRE(groups > 0 && groups <= 32 && c0 > 0 && c1 >= 0 && (c0 + c1) % groups == 0, "vd_gn_apply_sums_f16: C=%d must divide into at most 32 groups (groups=%d)", c0 + c1, groups);
    VD_REQUIRE((((size_t)sums0 | (size_t)sums1 | (size_t)gamma | (size_t)beta) & 15) == 0, "vd_gn_apply_sums_f16: sums / gamma / beta must be 16-byte aligned");
    GnApplyArgs a;
    a.table = nullptr;
    a.sums0 = reinterpret_cast<const long long*>(sums0); a.sums1 = reinterpret_cast<const long long*>(sums1);
    a.gamma = reinterpret_cast<const f16*>(gamma); a.beta = reinterpret_cast<const f16*>(beta); a.groups = groups; a.eps = eps;
    return gn_apply_launch("vd_gn_apply_sums_f16", a, x0, c0, x1, c1, B, HW, silu, out, stream);
}

static int gn_apply_launch(const char* who, GnApplyArgs& a, const void* x0, int c0, const void* x1, int c1, int B, int HW, int silu, void* out,
                           hipStream_t stream) {
    if (!x1) c1 = 0;
    VD_REQUIRE(B > 0 && B <= 65535 && HW > 0 && c0 > 0 && c0 % 8 == 0 && c1 % 8 == 0, "%s: channel counts must be multiples of 8", who);
    const int C = c0 + c1, C8 = C / 8;
    VD_REQUIRE(C8 <= 512, "%s: C=%d > 4096", who, C);
    a.x0 = reinterpret_cast<const f16*>(x0); a.x1 = reinterpret_cast<const f16*>(x1); a.out = reinterpret_cast<f16*>(out);
    a.c0 = c0; a.c1 = c1; a.HW = HW; a.silu = silu;
    a.TC = C8 < 256 ? C8 : 256;
    a.R = 256 / a.TC;
    a.npos = (C8 + a.TC - 1) / a.TC;
    // ~16K elements per block (the optimum the round-3 apply kernel measured), whole row-lane trips
    constexpr int chunk = 8192;   // elements per block: measured best with the light prologue (16384: +0.02 ms per forward, 32768: +0.12)
    int rpc = chunk / C;
    if (rpc < 1) rpc = 1;
    rpc = ((rpc + a.R - 1) / a.R) * a.R;
    if (rpc > HW) rpc = HW;
    a.rows_per_chunk = rpc;
    if (a.table) hipLaunchKernelGGL(gn_apply_table_kernel<false>, dim3((HW + rpc - 1) / rpc, B), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(gn_apply_table_kernel<true>, dim3((HW + rpc - 1) / rpc, B), dim3(256), 0, stream, a);
    return vd_check_launch(who);
}
