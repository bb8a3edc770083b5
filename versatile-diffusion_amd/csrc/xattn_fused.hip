// vd_xattn_f16: the query side of a cross-attention layer in ONE launch:
//
//     out[b, n, h*D : (h+1)*D] = softmax( (LayerNorm(x[b, n]) Wq_h^T) K_h^T * scale ) V_h
//
// Replaces, on the reference path, norm2 -> to_q -> rearrange -> einsum -> softmax -> einsum of
// /root/reference/lib/model_zoo/attention.py:170-193,216 (LayerNorm, Linear, two einsums around a softmax as separate torch
// ops), and in this library the chain vd_row_stats_f16 -> vd_gemm_f16(LN fold) -> vd_attention_f16: three launches that
// wrote and re-read the [M, C] query tensor and the row statistics for a context of 77 keys.  K / V of the context are
// step-invariant and come pre-projected (vd_gemm_f16 once per sample).
//
// Block = 4 waves = 128 query rows x ONE head.  Two phases in the same LDS:
//   1. projection  Q_h^T [D x 128] = Wq_h [D x C] . x^T [C x 128]  as a K loop over 64-channel chunks: the x chunk
//      [128 rows][64] and the head's weight chunk [32*DB rows][64] go global -> LDS by DMA (NST stages, counted vmcnt, one
//      barrier per chunk); MFMA operands A = weight rows, B = x rows, so a lane owns ONE query row (column l31) -- the
//      LayerNorm statistics of that row are accumulated from the very fragments the MFMA consumes (v_dot2: sum and sum of
//      squares, one lane^32 exchange at the end) and applied in the fold  q = rstd * (acc - mean * colsum) + bias
//      (gamma folded into Wq, beta into bias by the host, colsum = row sums of the folded Wq).
//      The A operand reads weight row swap23(l31) (bits 2 and 3 of the row index exchanged): accumulator register r of lane
//      (l31, hi) is then channel (r & 7) + 8 hi + 16 (r >> 3) -- registers 0..7 and 8..15 ARE the two B-operand fragments
//      (8 consecutive channels per lane half and 16-channel k-step) of the score MFMA: Q never leaves registers.
//   2. attention over the context tiles, exactly attn_fwd_kernel's loop (S^T = K Q^T swapped, online softmax with the
//      running max riding in the C operand, P from accumulator to B operand by a convert, V^T through ds_read_b64_tr_b16,
//      row sums through a column of ones).
// The H heads of one (batch, query block) run back to back on ONE XCD: the x tile is fetched into that L2 once.
#include <atomic>
#include <stdlib.h>
#include <type_traits>
#include "vd_common.h"
#include "../../include/vd_hip.h"

namespace {

constexpr int KV = 64;               // keys per tile
constexpr float RESCALE_THR = 6.0f;  // log2 units, see attention.hip

struct XAttnArgs {
    const f16* x;         // [B][Nq][C]
    const f16* wq;        // [C][C]   gamma-folded
    const f16* bq;        // [C]      beta-folded (Wq beta), may be null
    const float* colsum;  // [C]      row sums of the folded wq
    const f16* k;
    const f16* v;
    f16* o;               // [B][Nq][C]
    int H, Nq, Nk, C, ldk, ldv;
    int64_t sk, sv;
    float scale_log2, eps;
    int nqb, B, xcd_map;
};

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int xoff(int r, int s) { return r * 128 + ((s ^ ((r >> 1) & 7)) << 4); }   // [rows][64] f16, swizzled
__device__ __forceinline__ int swap23(int i) { return (i & ~12) | ((i & 4) << 1) | ((i & 8) >> 1); }

template <int N>
__device__ __forceinline__ void wait_vm_n() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// EK: the first K / V tile has its own LDS region and is requested before the projection starts (else both tiles alias the
// projection stages and are requested when the projection is done).  OCC: blocks per CU the register budget allows.
template <int D, int NST, bool EK, int OCC>
__global__ __launch_bounds__(256, OCC) void xattn_kernel(const XAttnArgs p) {
    constexpr int QB = 128;
    constexpr int KS = (D + 15) / 16;       // k-steps of the QK^T MFMA
    constexpr int DB = (D + 31) / 32;       // 32-row blocks of Q^T / O^T
    constexpr int WROWS = 32 * DB;
    constexpr int X_BYTES = QB * 128, W_BYTES = WROWS * 128, ST_BYTES = X_BYTES + W_BYTES;
    constexpr int PI = 4 + DB;              // LDS-DMA pieces per wave and chunk
    constexpr int CPR = 2 * KS + 1;         // K image, see attention.hip
    constexpr int KROW = CPR * 8;
    constexpr int K_BYTES = KV * CPR * 16;
    constexpr int V_BYTES = DB * KV * 64;
    constexpr int TILE_BYTES = K_BYTES + V_BYTES;
    constexpr int ALIAS_TILES = EK ? 1 : 2;   // K / V tile buffers that live in the projection stages
    constexpr int MAIN_BYTES = ((NST * ST_BYTES > ALIAS_TILES * TILE_BYTES) ? NST * ST_BYTES : ALIAS_TILES * TILE_BYTES) + (EK ? TILE_BYTES : 0);
    constexpr int TB0 = EK ? MAIN_BYTES - TILE_BYTES : 0;   // byte offset of tile buffer 0 / 1
    constexpr int TB1 = EK ? 0 : TILE_BYTES;
    constexpr int NQC = 16 * KS;            // channels of the (padded) head the fold touches
    constexpr bool HAS_ONES = (D % 32) != 0;
    constexpr int ONES_COL = D % 32;
    static_assert(D % 8 == 0 && NST >= 2 && NST <= 4, "instantiation");
    static_assert((NST - 2) * PI < 64, "vmcnt range");

    extern __shared__ __attribute__((aligned(1024))) char sm[];   // [NST stages | 2 K/V tiles] then colsum / bias of the head

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;

    int qb, b, h;
    {
        const int bid = blockIdx.x;
        int grp;
        if (p.xcd_map) {   // (B * nqb) % 8 == 0: the heads of a (batch, query block) group are consecutive on one XCD
            const int xcd = bid & 7, idx = bid >> 3;
            grp = xcd + 8 * (idx / p.H);
            h = idx % p.H;
        } else {
            grp = bid / p.H;
            h = bid % p.H;
        }
        b = grp / p.nqb;
        qb = grp % p.nqb;
    }
    const int C = p.C, NCH = C >> 6;
    const f16* kp = p.k + (size_t)b * p.sk + h * D;
    const f16* vp = p.v + (size_t)b * p.sv + h * D;
    f16* op = p.o + ((size_t)b * p.Nq) * C + h * D;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sm;
    constexpr unsigned OOB = 0x80000000u;

    // ---- K / V tile staging (attention.hip)
    const int qrow = qb * QB + wave * 32 + l31;
    const i32x4 rs_k = make_rsrc_words(kp, (unsigned)(((size_t)(p.Nk - 1) * p.ldk + D) * 2));
    const i32x4 rs_v = make_rsrc_words(vp, (unsigned)(((size_t)(p.Nk - 1) * p.ldv + D) * 2));
    constexpr int KM = (CPR + 3) / 4;
    unsigned voff_k[KM], voff_v[DB];
#pragma unroll
    for (int m = 0; m < KM; ++m) {
        const int q = (wave + 4 * m) * 64 + lane;
        const int r = q / CPR, slot = q - r * CPR;
        voff_k[m] = (slot * 8 < D) ? (unsigned)((r * p.ldk + slot * 8) * 2) : OOB;
    }
#pragma unroll
    for (int m = 0; m < DB; ++m) {
        const int q = (wave + 4 * m) * 64 + lane;
        const int d0 = (q >> 8) * 32 + (q & 3) * 8;
        voff_v[m] = (d0 < D) ? (unsigned)((((q >> 2) & 63) * p.ldv + d0) * 2) : OOB;
    }
    const unsigned k_tile_stride = (unsigned)(KV * p.ldk * 2), v_tile_stride = (unsigned)(KV * p.ldv * 2);
    auto stage = [&](int t, int buf) {
        const unsigned dst = lds0 + (unsigned)(buf ? TB1 : TB0) + (unsigned)(wave * 1024);
        const unsigned kt_off = (unsigned)t * k_tile_stride, vt_off = (unsigned)t * v_tile_stride;
#pragma unroll
        for (int m = 0; m < KM; ++m)
            if (m < CPR / 4 || wave < CPR % 4) dma16(rs_k, dst + m * 4096, voff_k[m] + kt_off, 0);
#pragma unroll
        for (int m = 0; m < DB; ++m) dma16(rs_v, dst + K_BYTES + m * 4096, voff_v[m] + vt_off, 0);
    };
    bool ones_lane[DB];
#pragma unroll
    for (int m = 0; m < DB; ++m) {
        const int q = (wave + 4 * m) * 64 + lane;
        ones_lane[m] = HAS_ONES && (q >> 8) == DB - 1 && (q & 3) == ONES_COL / 8;
    }
    auto plant_ones = [&](int buf) {
        if constexpr (HAS_ONES) {
#pragma unroll
            for (int m = 0; m < DB; ++m)
                if (ones_lane[m])
                    *reinterpret_cast<f16*>(sm + (buf ? TB1 : TB0) + K_BYTES + (wave + 4 * m) * 1024 + lane * 16) = (f16)1.0f;
        }
    };

    if constexpr (EK) stage(0, 0);   // oldest request of the block: landed long before the projection is done

    // ================================================================ phase 1: Q_h = LayerNorm(x) Wq_h^T
    const i32x4 rs_x = make_rsrc_words(p.x + ((size_t)b * p.Nq) * C, (unsigned)((size_t)p.Nq * C * 2));
    const i32x4 rs_w = make_rsrc_words(p.wq, (unsigned)((size_t)C * C * 2));
    unsigned voff_x[4], voff_w[DB];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int r = (wave + 4 * m) * 8 + (lane >> 3);
        const int grow = qb * QB + r;
        voff_x[m] = grow < p.Nq ? (unsigned)((grow * C + (((lane & 7) ^ ((r >> 1) & 7)) << 3)) * 2) : OOB;
    }
#pragma unroll
    for (int m = 0; m < DB; ++m) {
        const int r = (wave + 4 * m) * 8 + (lane >> 3);
        voff_w[m] = r < D ? (unsigned)(((h * D + r) * C + (((lane & 7) ^ ((r >> 1) & 7)) << 3)) * 2) : OOB;
    }
    auto issue = [&](int c, int s) {   // chunk c -> stage s
        const unsigned soff = (unsigned)(c * 128);
        const unsigned dst = lds0 + (unsigned)(s * ST_BYTES) + (unsigned)(wave * 1024);
#pragma unroll
        for (int m = 0; m < 4; ++m) dma16(rs_x, dst + m * 4096, voff_x[m], soff);
#pragma unroll
        for (int m = 0; m < DB; ++m) dma16(rs_w, dst + X_BYTES + m * 4096, voff_w[m], soff);
    };
#pragma unroll
    for (int c = 0; c < NST - 1; ++c)
        if (c < NCH) issue(c, c);

    // colsum / bias of the head's (padded) channels -> LDS, published by the first barrier of the loop
    float* cs_l = reinterpret_cast<float*>(sm + MAIN_BYTES);
    f16* bq_l = reinterpret_cast<f16*>(sm + MAIN_BYTES + NQC * 4);
    if (tid < NQC) {
        cs_l[tid] = tid < D ? p.colsum[h * D + tid] : 0.f;
        bq_l[tid] = (tid < D && p.bq) ? p.bq[h * D + tid] : (f16)0.f;
    }

    int rd_x[4], rd_w[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        rd_x[ks] = xoff(wave * 32 + l31, ks * 2 + hi);
        rd_w[ks] = X_BYTES + xoff(swap23(l31), ks * 2 + hi);   // + dt * 32 * 128 (the swizzle key repeats every 16 rows)
    }
    f32x16 qacc[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) qacc[i][r] = 0.f;
    float s1 = 0.f, s2 = 0.f;
    const f16x2 one2 = {(f16)1.0f, (f16)1.0f};

    int stg = 0;
    for (int c = 0; c < NCH; ++c) {
        // chunk c must have landed; the min(NST - 2, NCH - 1 - c) chunks issued after it may stay in flight
        if constexpr (NST == 2) {
            wait_vm_n<0>();
        } else {
            const int later = NCH - 1 - c;
            if (later >= NST - 2) wait_vm_n<(NST - 2) * PI>();
            else if (NST == 4 && later == 1) wait_vm_n<PI>();
            else wait_vm_n<0>();
        }
        __syncthreads();   // ... for every wave, and every wave has left chunk c - 1, whose stage is refilled now
        if (c + NST - 1 < NCH) issue(c + NST - 1, (stg + NST - 1) % NST);
        const char* st = sm + stg * ST_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            U4H8 xb;
            xb.u = *reinterpret_cast<const uint4*>(st + rd_x[ks]);
            {
                const f16x2* pr = reinterpret_cast<const f16x2*>(&xb);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s1 = __builtin_amdgcn_fdot2(pr[j], one2, s1, false);
                    s2 = __builtin_amdgcn_fdot2(pr[j], pr[j], s2, false);
                }
            }
#pragma unroll
            for (int dt = 0; dt < DB; ++dt) {
                U4H8 a;
                a.u = *reinterpret_cast<const uint4*>(st + rd_w[ks] + dt * 32 * 128);
                qacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, xb.h, qacc[dt], 0, 0, 0);
            }
        }
        stg = (stg + 1 == NST) ? 0 : stg + 1;
    }

    // LayerNorm fold + softmax scale -> the B-operand fragments of the score MFMA
    f16x8 qf[KS];
    {
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        const float inv_c = 1.0f / (float)C;
        const float mean = s1 * inv_c;
        float var = s2 * inv_c - mean * mean;
        if (var < 0.f) var = 0.f;
        const float rstd = rsqrtf(var + p.eps);
        const float a1 = rstd * p.scale_log2, a2 = -mean * a1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d0 = ks * 16 + hi * 8;
            const float4 c0 = *reinterpret_cast<const float4*>(cs_l + d0);
            const float4 c1 = *reinterpret_cast<const float4*>(cs_l + d0 + 4);
            U4H8 bb;
            bb.u = *reinterpret_cast<const uint4*>(bq_l + d0);
            const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float acc = qacc[ks >> 1][8 * (ks & 1) + e];
                qf[ks][e] = (f16)(fmaf(acc, a1, fmaf(cs[e], a2, (float)bb.e[e] * p.scale_log2)));
            }
        }
    }
    if constexpr (EK) plant_ones(0);   // (the last wait of the loop was vmcnt(0): this lane's own tile-0 pieces have landed)
    __syncthreads();   // every wave is done with the projection stages: the K / V tiles take their place

    // ================================================================ phase 2: attention over the context (attention.hip)
    f32x16 acc[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f32x16 negm;
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = 0.f;
    float l_run = 0.f;
    const int ntiles = (p.Nk + KV - 1) / KV;
    const unsigned v_lane = (unsigned)(K_BYTES + (4 * hi + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8);
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4* lds_h4_ptr;

    if constexpr (EK) {
        if (ntiles > 1) stage(1, 1);
    } else {
        stage(0, 0);
        if (ntiles > 1) {
            stage(1, 1);   // short contexts: both tiles requested at once
            wait_vm_n<CPR / 4 + DB>();   // at most the pieces of tile 1 (the fewest any wave issues) stay in flight
        } else {
            wait_vm_n<0>();
        }
        plant_ones(0);
        __syncthreads();
    }

    auto tile = [&](const int t, auto has_next) {
        constexpr bool HAS_NEXT = decltype(has_next)::value;
        const int cur = t & 1;
        const unsigned tbase = lds0 + (unsigned)(cur ? TB1 : TB0);
        const f16* Ks = reinterpret_cast<const f16*>(sm + (cur ? TB1 : TB0));

        f32x16 st[KV / 32];
#pragma unroll
        for (int kt = 0; kt < KV / 32; ++kt) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                U4H8 a;
                a.u = *reinterpret_cast<const uint4*>(Ks + (kt * 32 + l31) * KROW + ks * 16 + hi * 8);
                st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, qf[ks], ks == 0 ? negm : st[kt], 0, 0, 0);  // S' = s - m
            }
        }
        const int key0 = t * KV;
        if (key0 + KV > p.Nk) {
#pragma unroll
            for (int kt = 0; kt < KV / 32; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key >= p.Nk) st[kt][r] = -INFINITY;
                }
        }
        float mx = st[0][0];
#pragma unroll
        for (int kt = 0; kt < KV / 32; ++kt)
#pragma unroll
            for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, st[kt][r]), st[kt][r + 1]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        if (t == 0 || __any(mx > RESCALE_THR)) {
            float delta = (t == 0) ? mx : fmaxf(mx, 0.f);
            if (mx == -INFINITY) delta = 0.f;
            if (t != 0) {
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                if constexpr (!HAS_ONES) l_run *= alpha;
#pragma unroll
                for (int i = 0; i < DB; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] *= alpha;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) negm[r] -= delta;
#pragma unroll
            for (int kt = 0; kt < KV / 32; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[kt][r] -= delta;
        }
        f16x8 pb[KV / 32][2];
        if constexpr (HAS_ONES) {
#pragma unroll
            for (int kt = 0; kt < KV / 32; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) pb[kt][r >> 3][r & 7] = (f16)__builtin_amdgcn_exp2f(st[kt][r]);
        } else {
            float ps = 0.f;
#pragma unroll
            for (int kt = 0; kt < KV / 32; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f(st[kt][r]);
                    ps += e;
                    pb[kt][r >> 3][r & 7] = (f16)e;
                }
            l_run += ps;
        }
        const lds_h4_ptr vbase = (lds_h4_ptr)(size_t)(tbase + v_lane);
#pragma unroll
        for (int kt = 0; kt < KV / 32; ++kt)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int i = 0; i < DB; ++i) {
                    const int off8 = (i * KV * 64 + (kt * 32 + 16 * s) * 64) / 8;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(vbase + off8);
                    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(vbase + off8 + 64);
                    const f16x8 a = __builtin_bit_cast(f16x8, __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pb[kt][s], acc[i], 0, 0, 0);
                }
            }
        if constexpr (HAS_NEXT) {
            // tile t + 1 was requested one tile ago; tile t + 2 goes into the buffer every wave has left after the barrier
            wait_vm_n<0>();
            plant_ones(cur ^ 1);
            __syncthreads();
            if (t + 2 < ntiles) stage(t + 2, cur);
        }
    };
    for (int t = 0; t + 1 < ntiles; ++t) tile(t, std::true_type{});
    tile(ntiles - 1, std::false_type{});

    float l_tot;
    if constexpr (HAS_ONES) {
        constexpr int RL = (ONES_COL & 3) + 4 * (ONES_COL >> 3);
        static_assert(!HAS_ONES || (ONES_COL % 8 == 0 && ONES_COL < 32), "ones column must sit on a hi = 0 accumulator row");
        l_tot = __shfl(acc[DB - 1][RL], l31, 64);
    } else {
        l_tot = l_run + __shfl_xor(l_run, 32, 64);
    }
    const float inv = (l_tot > 0.f) ? 1.0f / l_tot : 0.f;
    if (qrow < p.Nq) {
#pragma unroll
        for (int i = 0; i < DB; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = i * 32 + 8 * g + 4 * hi;
                if (d0 < D) {
                    U2H4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) o.e[j] = (f16)(acc[i][g * 4 + j] * inv);
                    *reinterpret_cast<uint2*>(op + (size_t)qrow * C + d0) = o.u;
                }
            }
    }
}

template <int D, int NST, bool EK>
constexpr int xattn_lds_bytes() {
    constexpr int KS = (D + 15) / 16, DB = (D + 31) / 32;
    constexpr int ST = 128 * 128 + 32 * DB * 128;
    constexpr int TILE = KV * (2 * KS + 1) * 16 + DB * KV * 64;
    constexpr int AL = EK ? 1 : 2;
    return ((NST * ST > AL * TILE) ? NST * ST : AL * TILE) + (EK ? TILE : 0) + 16 * KS * 6;
}

template <int D, int NST, bool EK, int OCC>
int launch_xattn(XAttnArgs a, hipStream_t stream) {
    constexpr int LDS = xattn_lds_bytes<D, NST, EK>();
    static_assert(LDS * OCC <= 160 * 1024, "LDS budget at the requested occupancy");
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_kernel<D, NST, EK, OCC>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) {
            vd_set_error("vd_xattn_f16: cannot reserve %d bytes of LDS: %s", LDS, hipGetErrorString(e));
            return VD_ERR_LAUNCH;
        }
        done.fetch_or(bit, std::memory_order_release);
    }
    a.nqb = (a.Nq + 127) / 128;
    a.xcd_map = ((a.B * a.nqb) & 7) == 0 ? 1 : 0;
    hipLaunchKernelGGL((xattn_kernel<D, NST, EK, OCC>), dim3(a.nqb * a.B * a.H), dim3(256), LDS, stream, a);
    return vd_check_launch("vd_xattn_f16");
}

}  // namespace

extern "C" int vd_xattn_supported(int H, int D) {
    return (D == 40 || D == 80 || D == 160) && H > 0 && ((H * D) % 64) == 0 ? 1 : 0;
}

extern "C" int vd_xattn_f16(const void* x, const void* wq, const void* bq, const float* colsum, float ln_eps, const void* k,
                            const void* v, void* out, int B, int H, int Nq, int Nk, int D, int ldk, int ldv, int64_t sk,
                            int64_t sv, float scale, hipStream_t stream) {
    VD_REQUIRE(x && wq && colsum && k && v && out, "vd_xattn_f16: null pointer");
    VD_REQUIRE(B > 0 && H > 0 && Nq > 0 && Nk > 0, "vd_xattn_f16: empty problem B=%d H=%d Nq=%d Nk=%d", B, H, Nq, Nk);
    VD_REQUIRE(vd_xattn_supported(H, D), "vd_xattn_f16: unsupported heads x head dim %d x %d (head dim 40 / 80 / 160, width a multiple of 64)", H, D);
    VD_REQUIRE((ldk % 8 == 0) && (ldv % 8 == 0), "vd_xattn_f16: leading dimensions must keep 16-byte row alignment");
    VD_REQUIRE((((size_t)x | (size_t)wq | (size_t)k | (size_t)v | (size_t)out | (size_t)colsum) & 15) == 0 && (!bq || ((size_t)bq & 1) == 0),
               "vd_xattn_f16: operands must be 16-byte aligned");
    const int C = H * D;
    VD_REQUIRE((int64_t)Nq * C * 2 < (int64_t)1 << 31 && (int64_t)C * C * 2 < (int64_t)1 << 31,
               "vd_xattn_f16: one sample of x / the weight must stay below 2 GiB");
    VD_REQUIRE(((int64_t)Nk + KV) * ldk * 2 < (int64_t)1 << 31 && ((int64_t)Nk + KV) * ldv * 2 < (int64_t)1 << 31,
               "vd_xattn_f16: one (batch, head) K/V slice must stay below 2 GiB (Nk=%d ldk=%d ldv=%d)", Nk, ldk, ldv);
    XAttnArgs a;
    a.x = (const f16*)x; a.wq = (const f16*)wq; a.bq = (const f16*)bq; a.colsum = colsum;
    a.k = (const f16*)k; a.v = (const f16*)v; a.o = (f16*)out;
    a.H = H; a.Nq = Nq; a.Nk = Nk; a.C = C; a.ldk = ldk; a.ldv = ldv; a.sk = sk; a.sv = sv;
    a.scale_log2 = scale * 1.44269504088896340736f;
    a.eps = ln_eps;
    a.nqb = 0; a.B = B; a.xcd_map = 0;
    // development switch VD_XATTN_VAR: 1 = early K/V tile 0 (measured neutral: 42.6 / 24.3 / 32.7 us vs 43.3 / 24.1 / 30.4 at the
    // 64x64 / 32x32 / 16x16 levels), 2 = head dim 40 with the old 3-stage, 2-blocks-per-CU pipeline (43.3 us vs 38.3 at 3 blocks:
    // what hides the chunk latency is a third resident block, not a deeper pipeline)
    switch (D) {
        case 40: return launch_xattn<40, 2, false, 3>(a, stream);
        case 80: return launch_xattn<80, 2, false, 2>(a, stream);
        default: return launch_xattn<160, 4, false, 1>(a, stream);
    }
}
