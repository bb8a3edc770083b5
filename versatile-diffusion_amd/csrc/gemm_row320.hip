// vd_gemm_row320_f16: y[M][N] = [LayerNorm](x[M][320]) W[N][320]^T + bias (+ res) for the projections of the 64x64 level of the
// UNet (K = 320, N = 320 or 960: SpatialTransformer.proj_in / proj_out, CrossAttention.to_out and the LayerNorm-folded fused
// q | k | v projection; /root/reference/lib/model_zoo/attention.py:159-163,170-193,245-258).
//
// Why not gemm_f16_kernel: with K = 320 a tile has five K iterations, so prologue, epilogue and the activation DMA of every
// (row panel x column tile) block are most of its life (23 us for 6.7 GFLOP, 69 us for the N = 960 projection).  Here the
// ROWS are what a block keeps: its 128 rows of x live in registers as MFMA operand fragments for the whole launch
// (ff_geglu_kernel's scheme: a wave's 32 rows x 320 columns = 80 registers, layer-normalised in place when asked), so there is
// no activation tile in LDS, no activation DMA, no fragment read for it -- per MFMA one ds_read_b128 of the weight tile --
// and the [128 x 320] fp32 accumulator (80 registers) leaves through one LDS tile as 16-byte row segments (+ residual).
// Weights stream through three 40-KiB slots ([320 rows][64 k] as two 160-row halves; requested two K tiles ahead, counted
// vmcnt, one barrier per K tile).  MULTI (N = 960 and wider): ONE block walks all column groups of its rows -- x is loaded and
// normalised once, the weight tiles of consecutive groups form one continuous stream through the slots, and a group's
// accumulators leave through a small wave-private LDS patch (64 contiguous bytes per row and store) while the next group is
// already being multiplied.  The single-group kernel (blockIdx.y = column group, LDS-staged 16-byte row segments +
// residual) serves N = 320.
#include "gemm_kernel.h"

namespace {

constexpr int RG_C = 320;
constexpr int RG_KT = RG_C / 64;
constexpr int RG_BM = 128;
constexpr int RG_HALF = 160 * 128;            // bytes of one 160-row half of a K tile
constexpr int RG_SLOT = 2 * RG_HALF;          // 40 KiB
constexpr int RG_NSLOT = 3;
constexpr int RG_CS_LD = RG_C + 8;
constexpr int RG_LDS_MAIN = RG_NSLOT * RG_SLOT;
constexpr int RG_LDS_EPI = RG_BM * RG_CS_LD * 2;
constexpr int RG_LDS = RG_LDS_MAIN > RG_LDS_EPI ? RG_LDS_MAIN : RG_LDS_EPI;
constexpr int RG_MAXBIAS = 8 * 1024;        // MULTI: the bias vector parked behind the slots (N <= 4096)
constexpr int RG_PATCH = 32 * 80;            // MULTI: wave-private output patch [32 rows][32 columns], 80-byte pitch
constexpr int RG_MULTI_EXTRA = RG_MAXBIAS + 8 * RG_PATCH;
static_assert(RG_LDS + RG_MULTI_EXTRA <= 160 * 1024 && RG_LDS == RG_LDS_MAIN, "LDS budget");

struct RGArgs {
    const f16* x;      // [M][320]
    const f16* w;      // [N][320]   (gamma-folded when ln)
    const f16* bias;   // [N] or null (beta-folded when ln)
    const f16* res;    // [M][N] or null
    f16* y;            // [M][N]
    int M, N;
    float eps;
    int nt_store;
#ifdef VD_TIMELINE
    unsigned long long* tl;
#endif
};

template <bool LN, bool MULTI>
__global__ __launch_bounds__(512, 2) void rowgemm320_kernel(const RGArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    VD_TL_DECL;
    VD_TL(0);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int hi = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.x * RG_BM;
    const int n0 = MULTI ? 0 : blockIdx.y * RG_C;   // column group of 320 (MULTI: the block walks all of them)

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const i32x4 rs_w = make_rsrc_words(p.w + (size_t)n0 * RG_C, (unsigned)((MULTI ? p.N : RG_C) * RG_C * 2));

    // a 160-row half = 20 pieces of 8 rows: wave w issues pieces w, w + 8 and (w + 16 < 20 ? w + 16 : w + 8 again: identical
    // bytes to the identical place) -- every wave issues 6 pieces per K tile, so the waits can be counted
    unsigned v2[3], d2[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        int q = j * 8 + wave;
        if (q >= 20) q -= 8;
        const int r = q * 8 + (lane >> 3);
        v2[j] = (unsigned)((r * RG_C + (((lane & 7) ^ ((r >> 1) & 7)) << 3)) * 2);
        d2[j] = (unsigned)(__builtin_amdgcn_readfirstlane(q) * 1024);
    }
    auto issue_kt = [&](int kt, int slot, int grp = 0) {   // K tile kt of column group grp
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned soff = (unsigned)(((grp * RG_C + h * 160) * RG_C + kt * 64) * 2);
            const unsigned dst = lds0 + (unsigned)(slot * RG_SLOT + h * RG_HALF);
#pragma unroll
            for (int j = 0; j < 3; ++j) dma16(rs_w, dst + d2[j], v2[j], soff);
        }
    };

    // ---- x fragments: xf[kt * 4 + ks] = x[row][kt * 64 + ks * 16 + hi * 8 .. + 8] (B operand)
    const int row = m0 + wm * 32 + l31;
    const int rowc = row < p.M ? row : p.M - 1;
    f16x8 xf[RG_KT * 4];
    {
        const f16* xr = p.x + (size_t)rowc * RG_C + hi * 8;
#pragma unroll
        for (int k = 0; k < RG_KT * 4; ++k) {
            U4H8 t;
            t.u = *reinterpret_cast<const uint4*>(xr + k * 16);
            xf[k] = t.h;
        }
    }
    if constexpr (MULTI) {
        // bias -> LDS by DMA (read with ds_read in the group epilogues: no compiler-counted global load may sit between the
        // hand-counted requests of the loop -- hipcc would drain vmcnt, and with it the weight tiles in flight, in front of it)
        const int wave_s = __builtin_amdgcn_readfirstlane(wave);
        if (p.bias && wave_s * 512 < p.N) {
            const i32x4 rs_b = make_rsrc_words(p.bias, (unsigned)(p.N * 2));
            dma16(rs_b, lds0 + (unsigned)(RG_LDS_MAIN + wave_s * 1024), (unsigned)(wave_s * 1024 + lane * 16), 0u);
        }
    }
    issue_kt(0, 0);
    issue_kt(1, 1);
    issue_kt(2, 2);

    if constexpr (LN) {   // LayerNorm in registers: the two lanes l31 / l31 + 32 hold one row between them
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < RG_KT * 4; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) s += (float)xf[k][i];
        s += __shfl_xor(s, 32, 64);
        const float mean = s * (1.0f / RG_C);
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < RG_KT * 4; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float dl = (float)xf[k][i] - mean;
                q += dl * dl;
            }
        q += __shfl_xor(q, 32, 64);
        const float rstd = rsqrtf(q * (1.0f / RG_C) + p.eps);
        const float nmr = -mean * rstd;
#pragma unroll
        for (int k = 0; k < RG_KT * 4; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) xf[k][i] = (f16)fmaf((float)xf[k][i], rstd, nmr);
    } else {
        // the x loads are the only VMEM results the compiler tracks: consume them here so its vmcnt wait does not land inside
        // the loop (it cannot see the hand-counted DMA requests issued after them)
#pragma unroll
        for (int k = 0; k < RG_KT * 4; ++k) asm volatile("" ::"v"(xf[k]));
    }

    f32x16 acc[5];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    int rd[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) rd[ks] = wn * RG_HALF + lds_off_kb<64>(l31, ks * 2 + hi);   // + slot * RG_SLOT + j * 32 * 128

    if constexpr (MULTI) {
        // one continuous stream of K tiles over all column groups: tile ti = (group ti / 5, K tile ti % 5) sits in slot ti % 3,
        // tile ti + 2 is requested right after the barrier of tile ti
        const int ngrp = p.N / RG_C, ntile = ngrp * RG_KT;
        int slot = 0, grp = 0;
        for (int t0 = 0; t0 < ntile; t0 += RG_KT, ++grp) {
#pragma unroll
            for (int kt = 0; kt < RG_KT; ++kt) {
                const int ti = t0 + kt;
                // tile ti has landed.  What may still be in flight behind it: tile ti + 1 (6 requests; + tile ti + 2 at the very
                // first step), and around a group boundary the 10 output stores of the previous group's epilogue (stores count
                // in vmcnt on gfx9): they were issued behind tile ti + 1's requests and in front of tile ti + 2's
                if (ti == 0) wait_vm<12>();
                else if (ti + 1 >= ntile) wait_vm<0>();
                else if (kt <= 1 && grp > 0) wait_vm<16>();
                else wait_vm<6>();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (ti >= 1 && ti + 2 < ntile) {
                    const int tn = ti + 2;
                    issue_kt(tn % RG_KT, slot == 0 ? 2 : slot - 1, tn / RG_KT);   // (ti + 2) % 3: the slot tile ti - 1 has left
                }
                const char* st = smem + slot * RG_SLOT;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        U4H8 wf;
                        wf.u = *reinterpret_cast<const uint4*>(st + rd[ks] + j * 32 * 128);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf.h, xf[kt * 4 + ks], acc[j], 0, 0, 0);
                    }
                slot = slot == 2 ? 0 : slot + 1;
            }
            // the group's accumulators leave tile by tile through a wave-private LDS patch ([32 rows][32 columns], 80-byte row
            // pitch): + bias, fp16, then 16 bytes per lane = 64 contiguous bytes per row and store (10 stores per group: they
            // drain while the next group is multiplied).  Rows past M were loaded from row M - 1 and store its (identical)
            // values again: every wave issues exactly 10 stores, which the counted waits rely on.
            {
                char* patch = smem + RG_LDS_MAIN + RG_MAXBIAS + wave * RG_PATCH;
                const char* bl = smem + RG_LDS_MAIN + (grp * RG_C + wn * 160 + 4 * hi) * 2;
                const int prow = lane >> 2, pchunk = lane & 3;
#pragma unroll
                for (int j = 0; j < 5; ++j) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        U2H4 b, o;
                        b.u = make_uint2(0, 0);
                        if (p.bias) b.u = *reinterpret_cast<const uint2*>(bl + (j * 32 + 8 * g) * 2);
#pragma unroll
                        for (int q = 0; q < 4; ++q) o.e[q] = (f16)(acc[j][g * 4 + q] + (float)b.e[q]);
                        *reinterpret_cast<uint2*>(patch + l31 * 80 + (8 * g + 4 * hi) * 2) = o.u;
                    }
#pragma unroll
                    for (int it = 0; it < 2; ++it) {
                        const int r = prow + 16 * it;
                        const uint4 v = *reinterpret_cast<const uint4*>(patch + r * 80 + pchunk * 16);
                        int grow = m0 + wm * 32 + r;
                        if (grow >= p.M) grow = p.M - 1;
                        *reinterpret_cast<uint4*>(p.y + (size_t)grow * p.N + grp * RG_C + wn * 160 + j * 32 + pchunk * 8) = v;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 5; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        }
        return;
    } else {
#pragma unroll
        for (int kt = 0; kt < RG_KT; ++kt) {
            // K tile kt has landed; the tiles requested after it (kt + 1, and kt + 2 at the first step) may still be in flight
            if (kt == 0) wait_vm<12>();
            else if (kt + 1 < RG_KT) wait_vm<6>();
            else wait_vm<0>();
            __builtin_amdgcn_s_barrier();   // ... for every wave; every wave has left K tile kt - 1, whose slot is refilled now
            asm volatile("" ::: "memory");
#ifdef VD_TIMELINE
            if (kt == 0) VD_TL(1);
#endif
            if (kt >= 1 && kt + 2 < RG_KT) issue_kt(kt + 2, (kt + 2) % RG_NSLOT);
            const char* st = smem + (kt % RG_NSLOT) * RG_SLOT;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    U4H8 wf;
                    wf.u = *reinterpret_cast<const uint4*>(st + rd[ks] + j * 32 * 128);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf.h, xf[kt * 4 + ks], acc[j], 0, 0, 0);
                }
        }
    }
    __syncthreads();   // every wave is done with the slots: the output tile re-uses that LDS
    VD_TL(2);

    // ---- epilogue: + bias -> fp16 tile in LDS -> 16-byte row segments (+ residual) -> y
    f16* cs = reinterpret_cast<f16*>(smem);
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = wn * 160 + j * 32 + 8 * g + 4 * hi;
            U2H4 b, o;
            b.u = make_uint2(0, 0);
            if (p.bias) b.u = *reinterpret_cast<const uint2*>(p.bias + n0 + col);
#pragma unroll
            for (int q = 0; q < 4; ++q) o.e[q] = (f16)(acc[j][g * 4 + q] + (float)b.e[q]);
            *reinterpret_cast<uint2*>(cs + (wm * 32 + l31) * RG_CS_LD + col) = o.u;
        }
    __syncthreads();
    VD_TL(3);
    constexpr int CH = RG_C / 8;                       // 40 segments per row
    constexpr int PER = RG_BM * CH / 512;              // 10 per thread
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int sgm = tid + k * 512;
        const int r = sgm / CH, cc = (sgm % CH) * 8;
        const int grow = m0 + r;
        if (grow < p.M) {
            U4H8 t, o;
            t.u = *reinterpret_cast<const uint4*>(cs + r * RG_CS_LD + cc);
            o = t;
            if (p.res) {
                U4H8 a;
                a.u = *reinterpret_cast<const uint4*>(p.res + (size_t)grow * p.N + n0 + cc);
#pragma unroll
                for (int q = 0; q < 8; ++q) o.e[q] = (f16)((float)t.e[q] + (float)a.e[q]);
            }
            f16* dst = p.y + (size_t)grow * p.N + n0 + cc;
            if (p.nt_store) vd_store16_nt(dst, o.u);
            else *reinterpret_cast<uint4*>(dst) = o.u;
        }
    }
    VD_TL(4);
    VD_TL_FLUSH(p.tl);
}

// ---- vd_gemm_row320_chain_f16: GroupNorm (as a per-sample affine map) -> proj_in -> h, then LayerNorm(h) -> q | k | v, in ONE
// launch for the entry of a SpatialTransformer at inner width 320 (/root/reference/lib/model_zoo/attention.py:236-258 norm +
// proj_in, :214 norm1, :170-176 to_q / to_k / to_v).  The MULTI scheme with one more column group in front: group 0 multiplies
// the affine-mapped rows of x with the proj_in weights; its output h leaves as 64-byte row segments (it is the residual
// stream), the block waits for its own stores, and every wave reads the 320 columns of ITS rows back as operand fragments
// (the two waves of a row block hold 160 columns each), layer-normalises them in registers and walks the q | k | v groups.
// The weight tiles of all four groups form one stream through the slots.
struct RCArgs {
    const f16* x;       // [M][320]
    const f16* sc;      // [M / rows_per_image][320]  GroupNorm scale  (rstd * gamma)
    const f16* sh;      // [M / rows_per_image][320]  GroupNorm shift  (beta - mean * scale; with `ctr`: beta - (mean - ctr) * scale)
    const f16* ctr;     // [M / rows_per_image][320]  fp16(mean) per channel or null: x is centred before the scale (no |mean| / sigma loss)
    const f16* w1;      // [320][320]  proj_in
    const f16* b1;      // [320]
    f16* h;             // [M][320]
    const f16* w2;      // [N2][320]   gamma-folded q | k | v
    const f16* b2;      // [N2] or null (beta-folded)
    f16* y2;            // [M][N2]
    int M, N2, rows_per_image;
    float eps;
};

__global__ __launch_bounds__(512, 2) void rowchain320_kernel(const RCArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave >> 1, wn = wave & 1;
    const int hi = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.x * RG_BM;

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const i32x4 rs_w1 = make_rsrc_words(p.w1, (unsigned)(RG_C * RG_C * 2));
    const i32x4 rs_w2 = make_rsrc_words(p.w2, (unsigned)(p.N2 * RG_C * 2));
    unsigned v2[3], d2[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        int q = j * 8 + wave;
        if (q >= 20) q -= 8;
        const int r = q * 8 + (lane >> 3);
        v2[j] = (unsigned)((r * RG_C + (((lane & 7) ^ ((r >> 1) & 7)) << 3)) * 2);
        d2[j] = (unsigned)(__builtin_amdgcn_readfirstlane(q) * 1024);
    }
    auto issue_tile = [&](int ti, int slot) {   // tile ti: group ti / 5 (0 = proj_in, g >= 1 = column group g - 1 of w2), K tile ti % 5
        const int grp = ti / RG_KT, kt = ti - grp * RG_KT;
        const i32x4 rs = grp == 0 ? rs_w1 : rs_w2;
        const int g2 = grp == 0 ? 0 : grp - 1;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned soff = (unsigned)(((g2 * RG_C + h * 160) * RG_C + kt * 64) * 2);
            const unsigned dst = lds0 + (unsigned)(slot * RG_SLOT + h * RG_HALF);
#pragma unroll
            for (int j = 0; j < 3; ++j) dma16(rs, dst + d2[j], v2[j], soff);
        }
    };

    const int row = m0 + wm * 32 + l31;
    const int rowc = row < p.M ? row : p.M - 1;
    f16x8 xf[RG_KT * 4];
    {
        const f16* xr = p.x + (size_t)rowc * RG_C + hi * 8;
#pragma unroll
        for (int k = 0; k < RG_KT * 4; ++k) {
            U4H8 t;
            t.u = *reinterpret_cast<const uint4*>(xr + k * 16);
            xf[k] = t.h;
        }
    }
    // biases -> LDS by DMA: b1 at [0, 640), b2 at [1024, ...)
    if (wave_s == 0) {
        const i32x4 rs_b = make_rsrc_words(p.b1, (unsigned)(RG_C * 2));
        dma16(rs_b, lds0 + (unsigned)RG_LDS_MAIN, (unsigned)(lane * 16), 0u);
    } else if (p.b2 && (wave_s - 1) * 512 < p.N2) {
        const i32x4 rs_b = make_rsrc_words(p.b2, (unsigned)(p.N2 * 2));
        dma16(rs_b, lds0 + (unsigned)(RG_LDS_MAIN + 1024 + (wave_s - 1) * 1024), (unsigned)((wave_s - 1) * 1024 + lane * 16), 0u);
    }
    issue_tile(0, 0);
    issue_tile(1, 1);
    issue_tile(2, 2);

    {   // GroupNorm as an affine map of the sample (all 128 rows of a block belong to one image): x * scale + shift, packed fp16
        const int img = m0 / p.rows_per_image;
        const f16* scp = p.sc + (size_t)img * RG_C + hi * 8;
        const f16* shp = p.sh + (size_t)img * RG_C + hi * 8;
        if (p.ctr != nullptr) {   // (uniform) centred form: (x - fp16(mean)) is exact near the mean, scale and shift' are O(1)
            const f16* ctp = p.ctr + (size_t)img * RG_C + hi * 8;
#pragma unroll
            for (int k = 0; k < RG_KT * 4; ++k) {
                U4H8 a, b, c;
                a.u = *reinterpret_cast<const uint4*>(scp + k * 16);
                b.u = *reinterpret_cast<const uint4*>(shp + k * 16);
                c.u = *reinterpret_cast<const uint4*>(ctp + k * 16);
                xf[k] = (xf[k] - c.h) * a.h + b.h;
            }
        } else {
#pragma unroll
            for (int k = 0; k < RG_KT * 4; ++k) {
                U4H8 a, b;
                a.u = *reinterpret_cast<const uint4*>(scp + k * 16);
                b.u = *reinterpret_cast<const uint4*>(shp + k * 16);
                xf[k] = xf[k] * a.h + b.h;
            }
        }
    }

    f32x16 acc[5];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    int rd[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) rd[ks] = wn * RG_HALF + lds_off_kb<64>(l31, ks * 2 + hi);

    char* patch = smem + RG_LDS_MAIN + RG_MAXBIAS + wave * RG_PATCH;
    const int prow = lane >> 2, pchunk = lane & 3;
    const int ngrp = 1 + p.N2 / RG_C, ntile = ngrp * RG_KT;
    int slot = 0, grp = 0;
    for (int t0 = 0; t0 < ntile; t0 += RG_KT, ++grp) {
#pragma unroll
        for (int kt = 0; kt < RG_KT; ++kt) {
            const int ti = t0 + kt;
            if (ti == 0) wait_vm<12>();
            else if (ti + 1 >= ntile) wait_vm<0>();
            else if (kt <= 1 && grp > 0) wait_vm<16>();   // (after group 0 everything has been drained: 16 is merely permissive)
            else wait_vm<6>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (ti >= 1 && ti + 2 < ntile) issue_tile(ti + 2, slot == 0 ? 2 : slot - 1);
            const char* st = smem + slot * RG_SLOT;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    U4H8 wf;
                    wf.u = *reinterpret_cast<const uint4*>(st + rd[ks] + j * 32 * 128);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf.h, xf[kt * 4 + ks], acc[j], 0, 0, 0);
                }
            slot = slot == 2 ? 0 : slot + 1;
        }
        // ---- the group's accumulators leave tile by tile through the wave's LDS patch (see rowgemm320_kernel MULTI)
        {
            const bool first = grp == 0;
            const char* bl = smem + RG_LDS_MAIN + (first ? 0 : 1024 + (grp - 1) * RG_C * 2) + (wn * 160 + 4 * hi) * 2;
            const bool has_bias = first || p.b2 != nullptr;
            f16* ybase = first ? p.h : p.y2 + (grp - 1) * RG_C;
            const int ld = first ? RG_C : p.N2;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    U2H4 b, o;
                    b.u = make_uint2(0, 0);
                    if (has_bias) b.u = *reinterpret_cast<const uint2*>(bl + (j * 32 + 8 * g) * 2);
#pragma unroll
                    for (int q = 0; q < 4; ++q) o.e[q] = (f16)(acc[j][g * 4 + q] + (float)b.e[q]);
                    *reinterpret_cast<uint2*>(patch + l31 * 80 + (8 * g + 4 * hi) * 2) = o.u;
                }
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    const int r = prow + 16 * it;
                    const uint4 v = *reinterpret_cast<const uint4*>(patch + r * 80 + pchunk * 16);
                    int grow = m0 + wm * 32 + r;
                    if (grow >= p.M) grow = p.M - 1;
                    *reinterpret_cast<uint4*>(ybase + (size_t)grow * ld + wn * 160 + j * 32 + pchunk * 8) = v;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        if (grp == 0) {
            // h is complete in memory once every wave's stores are acknowledged: read this lane's row back as operand
            // fragments (all 320 columns: half of them were written by the partner wave) and layer-normalise in registers
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const f16* hr = p.h + (size_t)rowc * RG_C + hi * 8;
#pragma unroll
            for (int k = 0; k < RG_KT * 4; ++k) {
                typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
                const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(hr + k * 16));   // past the L1
                xf[k] = __builtin_bit_cast(f16x8, v);
            }
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < RG_KT * 4; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) s += (float)xf[k][i];
            s += __shfl_xor(s, 32, 64);
            const float mean = s * (1.0f / RG_C);
            float q = 0.f;
#pragma unroll
            for (int k = 0; k < RG_KT * 4; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float dl = (float)xf[k][i] - mean;
                    q += dl * dl;
                }
            q += __shfl_xor(q, 32, 64);
            const float rstd = rsqrtf(q * (1.0f / RG_C) + p.eps);
            const float nmr = -mean * rstd;
#pragma unroll
            for (int k = 0; k < RG_KT * 4; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) xf[k][i] = (f16)fmaf((float)xf[k][i], rstd, nmr);
        }
    }
}

template <bool LN, bool MULTI>
int launch_rowgemm(const RGArgs& a, hipStream_t stream) {
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rowgemm320_kernel<LN, MULTI>), hipFuncAttributeMaxDynamicSharedMemorySize, RG_LDS + (MULTI ? RG_MULTI_EXTRA : 0));
        if (e != hipSuccess) {
            vd_set_error("vd_gemm_row320_f16: cannot reserve %d bytes of LDS: %s", RG_LDS, hipGetErrorString(e));
            return VD_ERR_LAUNCH;
        }
        done.fetch_or(bit, std::memory_order_release);
    }
    const dim3 grid((unsigned)((a.M + RG_BM - 1) / RG_BM), MULTI ? 1u : (unsigned)(a.N / RG_C));
    constexpr int LDS = RG_LDS + (MULTI ? RG_MULTI_EXTRA : 0);
    hipLaunchKernelGGL((rowgemm320_kernel<LN, MULTI>), grid, dim3(512), LDS, stream, a);
    return vd_check_launch("vd_gemm_row320_f16");
}

}  // namespace

#ifdef VD_TIMELINE
static unsigned long long* g_timeline_row = nullptr;
extern "C" void vd_debug_set_timeline_row320(void* buf) { g_timeline_row = reinterpret_cast<unsigned long long*>(buf); }
#endif

extern "C" int vd_gemm_row320_supported(int64_t M, int N, int K) {
    return (K == RG_C && N > 0 && N % RG_C == 0 && M > 0 && M < (1ll << 31) / N) ? 1 : 0;
}

extern "C" int vd_gemm_row320_f16(const void* x, const void* w, const void* bias, const void* res, void* y, int64_t M, int N,
                                  int layernorm, float ln_eps, hipStream_t stream) {
    VD_REQUIRE(x && w && y, "vd_gemm_row320_f16: null pointer");
    VD_REQUIRE(vd_gemm_row320_supported(M, N, RG_C), "vd_gemm_row320_f16: M=%ld N=%d not supported (K = 320, N a multiple of 320)", (long)M, N);
    VD_REQUIRE((((size_t)x | (size_t)w | (size_t)y | (size_t)res) & 15) == 0 && ((size_t)bias & 7) == 0,
               "vd_gemm_row320_f16: operands must be 16-byte aligned (bias 8)");
    RGArgs a;
    a.x = (const f16*)x; a.w = (const f16*)w; a.bias = (const f16*)bias; a.res = (const f16*)res; a.y = (f16*)y;
    a.M = (int)M; a.N = N; a.eps = ln_eps;
#ifdef VD_TIMELINE
    a.tl = g_timeline_row;
#endif
    a.nt_store = 1;   // non-temporal stores of write-once outputs
    // more than one column group and no residual: one block per row block walks them all
    const bool multi = N > RG_C && N <= 4096 && res == nullptr;
    if (multi) return layernorm ? launch_rowgemm<true, true>(a, stream) : launch_rowgemm<false, true>(a, stream);
    return layernorm ? launch_rowgemm<true, false>(a, stream) : launch_rowgemm<false, false>(a, stream);
}

extern "C" int vd_gemm_row320_chain_f16(const void* x, const void* gn_scale, const void* gn_shift, const void* gn_center, int rows_per_image,
                                        const void* w1, const void* b1, void* h, const void* w2, const void* b2, void* y2,
                                        int64_t M, int N2, float ln_eps, hipStream_t stream) {
    VD_REQUIRE(x && gn_scale && gn_shift && w1 && b1 && h && w2 && y2, "vd_gemm_row320_chain_f16: null pointer");
    VD_REQUIRE(M > 0 && N2 > 0 && N2 % RG_C == 0 && N2 <= 3072 && M < (1ll << 31) / N2, "vd_gemm_row320_chain_f16: M=%ld N2=%d not supported", (long)M, N2);
    VD_REQUIRE(rows_per_image > 0 && rows_per_image % RG_BM == 0 && M % rows_per_image == 0,
               "vd_gemm_row320_chain_f16: rows per image (%d) must be a multiple of %d and divide M", rows_per_image, RG_BM);
    VD_REQUIRE((((size_t)x | (size_t)gn_scale | (size_t)gn_shift | (size_t)gn_center | (size_t)w1 | (size_t)b1 | (size_t)h | (size_t)w2 | (size_t)b2 | (size_t)y2) & 15) == 0,
               "vd_gemm_row320_chain_f16: operands must be 16-byte aligned");
    constexpr int LDS = RG_LDS + RG_MULTI_EXTRA;
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rowchain320_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) {
            vd_set_error("vd_gemm_row320_chain_f16: cannot reserve %d bytes of LDS: %s", LDS, hipGetErrorString(e));
            return VD_ERR_LAUNCH;
        }
        done.fetch_or(bit, std::memory_order_release);
    }
    RCArgs a;
    a.x = (const f16*)x; a.sc = (const f16*)gn_scale; a.sh = (const f16*)gn_shift; a.ctr = (const f16*)gn_center; a.w1 = (const f16*)w1; a.b1 = (const f16*)b1;
    a.h = (f16*)h; a.w2 = (const f16*)w2; a.b2 = (const f16*)b2; a.y2 = (f16*)y2;
    a.M = (int)M; a.N2 = N2; a.rows_per_image = rows_per_image; a.eps = ln_eps;
    hipLaunchKernelGGL(rowchain320_kernel, dim3((unsigned)((M + RG_BM - 1) / RG_BM)), dim3(512), LDS, stream, a);
    return vd_check_launch("vd_gemm_row320_chain_f16");
}
