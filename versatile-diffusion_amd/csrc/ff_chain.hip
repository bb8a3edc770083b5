// vd_ff_chain_f16 (round 5): the row-local tail of a 64x64-level transformer block in ONE launch, C = 320:
//
//     x1  = a Wo^T + bo + x0                                       (PRE:  attn2.to_out + residual;  a = cross-attention output)
//     y   = x1 + ( v (*) gelu(g) ) W2^T + b2,  [v | g] = LayerNorm(x1) W1^T + b1      (the gated feed-forward, as vd_ff_geglu_f16)
//     out = alpha ( y Wp^T + bp ) + r                              (POST: SpatialTransformer.proj_out + skip / context mixing)
//
// Replaces, on the reference path, attn2.to_out -> "+ x" -> norm3 -> ff -> "+ x" -> proj_out -> "+ x_in" of
// /root/reference/lib/model_zoo/attention.py:37-64,192-193,214-218,262-266 and, in this library, the launches
// vd_gemm_f16 (to_out, 27 us) -> vd_ff_geglu_f16 (107 us) -> vd_gemm_f16 (proj_out, 27 us): the two C x C projections of a
// 128-row block are 100 MFMAs per wave next to the 1200 of the feed-forward, but as launches of their own each reads and writes
// the whole [M, 320] activation once more (63 MB of traffic for 6.7 GFLOP) and pays a launch.
//
// The feed-forward main loop is ff_geglu_kernel's (ff_fused.hip: x rows as MFMA operand fragments in registers, seven weight
// slots with a static assignment, hand-counted vmcnt).  A projection stage works on the block's [128 x 320] fp16 tile in LDS
// (the epilogue tile of ff_geglu_kernel, pitch 328 halfs: conflict-free ds_read_b128 of operand fragments): its weights stream
// in ten 32-deep K tiles (320 rows x 64 bytes = 20 KiB) through a ring of three slots behind the tile, requested two tiles
// ahead, one barrier per tile; the result goes back into the tile through the registers (bias, alpha), then leaves as 16-byte
// row segments with the residual added -- and, for POST, the per-channel statistics of the stored tile for the GroupNorm of the
// next ResBlock (emit_chan_stats).  x1 is parked in global memory (scratch [M, 320]) between PRE and the final residual.
#include "gemm_kernel.h"

namespace {

constexpr int FC_C = 320;
constexpr int FC_KT = FC_C / 64;
constexpr int FC_HID = 4 * FC_C;
constexpr int FC_NCH = FC_HID / 64;
constexpr int FC_BM = 128;
constexpr int FC_S1 = 16384;
constexpr int FC_S2 = 20480;
constexpr int FC_W2 = FC_KT * FC_S1;
constexpr int FC_HT = FC_W2 + 2 * FC_S2;
constexpr int FC_B1 = FC_HT + 2 * FC_BM * 128;
constexpr int FC_LDS_MAIN = FC_B1 + 2 * FC_HID * 2;
constexpr int FC_CS_LD = FC_C + 8;
constexpr int FC_TILE = FC_BM * FC_CS_LD * 2;           // 83,968 bytes: the [128 x 320] fp16 tile
constexpr int FC_PS = 320 * 64;                         // one 32-deep K tile of a projection: 320 rows x 64 bytes
constexpr int FC_RING = FC_TILE;                        // three slots behind the tile
constexpr int FC_LDS = FC_LDS_MAIN > FC_RING + 3 * FC_PS ? FC_LDS_MAIN : FC_RING + 3 * FC_PS;
static_assert(FC_LDS <= 160 * 1024, "LDS budget");
static_assert(FC_RING + stat_lanes(FC_C) * FC_C * 8 <= FC_LDS, "statistics scratch behind the tile");

struct FCArgs {
    const f16* x;      // [M][C]  input of the feed-forward (PRE: the residual x0 of the first projection)
    const f16* a;      // PRE: [M][C] operand of the first projection
    const f16* wo;     // PRE: [C][C]
    const f16* bo;     // PRE: [C]
    f16* x1;           // PRE: [M][C] scratch, x1 parked for the final residual
    const f16* w1;     // [8C][C]  gamma-folded, GEGLU-packed
    const f16* b1;     // [8C]
    const f16* w2;     // [C][4C]
    const f16* b2;     // [C]
    const f16* wp;     // POST: [C][C]
    const f16* bp;     // POST: [C]
    const f16* r;      // POST: [M][C] residual of the last projection
    f16* y;            // [M][C] output (POST: out, else y)
    float* out_stats;  // POST: fp32 [M / 128][C][2] per-channel (mean, M2) of the stored rows, or null
    unsigned long long* stat_sums;   // POST, with out_stats: int64 [M / img_rows][C][2] fixed-point sums (VdGemmDesc.stat_sums), or null
    int img_rows;
    int M;
    float eps, alpha;
    int nt_store;
};

template <bool PRE, bool POST>
__global__ __launch_bounds__(512, 2) void ff_chain_kernel(const FCArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave >> 1, wn = wave & 1;
    const int hi = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.x * FC_BM;

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const i32x4 rs_w1 = make_rsrc_words(p.w1, (unsigned)(2 * FC_HID * FC_C * 2));
    const i32x4 rs_w2 = make_rsrc_words(p.w2, (unsigned)(FC_C * FC_HID * 2));
    f16* cs = reinterpret_cast<f16*>(smem);

    f32x16 acc2[5];
    auto zero_acc2 = [&]() {
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;
    };

    // ---- a C x C projection of the tile in LDS: acc2 (this wave's 32 rows x 160 columns, transposed) += tile W^T.
    // K tile t = columns 32 t .. 32 t + 31 of W: 20 pieces of 16 rows, wave w requests pieces w, w + 8 and (w + 16 < 20 ? w + 16 :
    // w + 8 again); lane -> row 16 q + lane / 4, physical 16-byte slot lane % 4 holding logical slot (lane % 4) ^ swz32(row).
    unsigned pv[3], pd[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        int q = j * 8 + wave;
        if (q >= 20) q -= 8;
        const int r = q * 16 + (lane >> 2);
        pv[j] = (unsigned)((r * FC_C + (((lane & 3) ^ lds_swz<32>(r)) << 3)) * 2);
        pd[j] = (unsigned)(__builtin_amdgcn_readfirstlane(q) * 1024);
    }
    int prd[2];   // weight fragment of k-step ks: row wn * 160 + j * 32 + l31 (+ j * 32 * 64 bytes), slot 2 ks + hi
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) prd[ks] = lds_off_kb<32>(wn * 160 + l31, ks * 2 + hi);
    const int brd = (wm * 32 + l31) * (FC_CS_LD * 2) + hi * 16;   // + (32 t + 16 ks) * 2: operand fragment of the tile row
    auto proj_stage = [&](const f16* W) {
        const i32x4 rs = make_rsrc_words(W, (unsigned)(FC_C * FC_C * 2));
        auto issue = [&](int t) {   // K tile t -> ring slot t % 3
            const unsigned dst = lds0 + (unsigned)(FC_RING + (t % 3) * FC_PS);
#pragma unroll
            for (int j = 0; j < 3; ++j) dma16(rs, dst + pd[j], pv[j], (unsigned)(t * 64));
        };
        issue(0);
        issue(1);
#pragma unroll
        for (int t = 0; t < 10; ++t) {
            if (t + 1 < 10) wait_vm<3>();   // tile t has landed for this wave (tile t + 1's three pieces may be in flight)
            else wait_vm<0>();
            __builtin_amdgcn_s_barrier();   // ... for every wave; every wave has left tile t - 1, whose slot is refilled below
            asm volatile("" ::: "memory");
            if (t + 2 < 10) issue(t + 2);
            const char* st = smem + FC_RING + (t % 3) * FC_PS;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                U4H8 bf;
                bf.u = *reinterpret_cast<const uint4*>(smem + brd + (32 * t + 16 * ks) * 2);
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    U4H8 wf;
                    wf.u = *reinterpret_cast<const uint4*>(st + prd[ks] + j * 32 * 64);
                    acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf.h, bf.h, acc2[j], 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_s_barrier();   // every wave is done with the tile and the ring
        asm volatile("" ::: "memory");
    };
    // registers -> tile: (acc2 + bias) * scale as fp16 at (row wm * 32 + l31, column wn * 160 + j * 32 + 8 g + 4 hi + q)
    auto acc_to_tile = [&](const f16* bias, float scale) {
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = wn * 160 + j * 32 + 8 * g + 4 * hi;
                U2H4 b, o;
                b.u = *reinterpret_cast<const uint2*>(bias + col);
#pragma unroll
                for (int q = 0; q < 4; ++q) o.e[q] = (f16)((acc2[j][g * 4 + q] + (float)b.e[q]) * scale);
                *reinterpret_cast<uint2*>(cs + (wm * 32 + l31) * FC_CS_LD + col) = o.u;
            }
    };
    constexpr int CH = FC_C / 8;                       // 40 segments per row
    constexpr int PER = FC_BM * CH / 512;              // 10 per thread
    // 16-byte row segments of the tile: tile + add (global) -> stored to dst (STORE) and / or written back into the tile (KEEP).
    // seg_request issues the loads of the add operand for all of a thread's segments back to back; seg_finish consumes them.  Round 6: the
    // two projections request their residual IN FRONT of the K loop (the ten registers wait across it), so its round trip -- ~2 us with the
    // whole chip asking at once -- no longer sits between the projection and the next stage.
    struct SegAdd { uint4 v[PER]; };
    auto seg_request = [&](const f16* add) {
        // (the thread index is laundered: hipcc would otherwise keep the ten 64-bit segment addresses of every call alive across
        // the feed-forward loop -- 60-80 spilled registers -- instead of recomputing them)
        int tl = tid;
        asm volatile("" : "+v"(tl));
        SegAdd a;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int sgm = tl + k * 512;
            const int r = sgm / CH, cc = (sgm % CH) * 8;
            const int grow = m0 + r < p.M ? m0 + r : p.M - 1;
            a.v[k] = *reinterpret_cast<const uint4*>(add + (size_t)grow * FC_C + cc);
        }
        return a;
    };
    auto seg_finish = [&](auto store_tag, auto keep_tag, const SegAdd& ad, f16* dst, bool nt) {
        constexpr bool STORE = decltype(store_tag)::value, KEEP = decltype(keep_tag)::value;
        int tl = tid;
        asm volatile("" : "+v"(tl));
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int sgm = tl + k * 512;
            const int r = sgm / CH, cc = (sgm % CH) * 8;
            const int grow = m0 + r;
            U4H8 t, a, o;
            t.u = *reinterpret_cast<const uint4*>(cs + r * FC_CS_LD + cc);
            a.u = ad.v[k];
#pragma unroll
            for (int q = 0; q < 8; ++q) o.e[q] = (f16)((float)t.e[q] + (float)a.e[q]);
            if constexpr (STORE) {
                if (grow < p.M) {
                    f16* d = dst + (size_t)grow * FC_C + cc;
                    if (nt) vd_store16_nt(d, o.u);
                    else *reinterpret_cast<uint4*>(d) = o.u;
                }
            }
            if constexpr (KEEP) *reinterpret_cast<uint4*>(cs + r * FC_CS_LD + cc) = o.u;
        }
    };
    auto segments = [&](auto store_tag, auto keep_tag, const f16* add, f16* dst, bool nt) {
        const SegAdd ad = seg_request(add);
        seg_finish(store_tag, keep_tag, ad, dst, nt);
    };
    constexpr std::true_type YES{};
    constexpr std::false_type NO{};

    const int row = m0 + wm * 32 + l31;
    const int rowc = row < p.M ? row : p.M - 1;
    f16x8 xf[FC_KT * 4];
    if constexpr (PRE) {
        // ---- x1 = a Wo^T + bo + x0: a -> tile, projection, (+ bo) -> tile, + x0 -> tile and scratch, operand fragments from the tile
        {
            int tl = tid;
            asm volatile("" : "+v"(tl));
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int sgm = tl + k * 512;
                const int r = sgm / CH, cc = (sgm % CH) * 8;
                const int grow = m0 + r < p.M ? m0 + r : p.M - 1;
                *reinterpret_cast<uint4*>(cs + r * FC_CS_LD + cc) = *reinterpret_cast<const uint4*>(p.a + (size_t)grow * FC_C + cc);
            }
        }
        zero_acc2();
        wait_vm<0>();
        __syncthreads();
        const SegAdd x0seg = seg_request(p.x);   // the residual of this projection: in flight across its K loop
        proj_stage(p.wo);
        acc_to_tile(p.bo, 1.0f);
        wait_vm<0>();
        __syncthreads();
        seg_finish(YES, YES, x0seg, p.x1, false);
        wait_vm<0>();
        __syncthreads();
#pragma unroll
        for (int k = 0; k < FC_KT * 4; ++k) {
            U4H8 t;
            t.u = *reinterpret_cast<const uint4*>(smem + (wm * 32 + l31) * (FC_CS_LD * 2) + (k * 16 + hi * 8) * 2);
            xf[k] = t.h;
        }
        __syncthreads();   // the tile has been read: the weight slots of the feed-forward may overwrite it
    } else {
        const f16* xr = p.x + (size_t)rowc * FC_C + hi * 8;
#pragma unroll
        for (int k = 0; k < FC_KT * 4; ++k) {
            U4H8 t;
            t.u = *reinterpret_cast<const uint4*>(xr + k * 16);
            xf[k] = t.h;
        }
    }

    // ================= the feed-forward: ff_geglu_kernel<1> (ff_fused.hip), unchanged =================
    unsigned v1[2], v2[3], d2[3];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (j * 8 + wave) * 8 + (lane >> 3);
        v1[j] = (unsigned)((r * FC_C + (((lane & 7) ^ ((r >> 1) & 7)) << 3)) * 2);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        int q = j * 8 + wave;
        if (q >= 20) q -= 8;
        const int r = q * 8 + (lane >> 3);
        v2[j] = (unsigned)((r * FC_HID + (((lane & 7) ^ ((r >> 1) & 7)) << 3)) * 2);
        d2[j] = (unsigned)(__builtin_amdgcn_readfirstlane(q) * 1024);
    }
    auto issue_w1 = [&](int hc, int kt) {
        const unsigned soff = (unsigned)((hc * 128 * FC_C + kt * 64) * 2);
        const unsigned dst = lds0 + (unsigned)(kt * FC_S1 + wave_s * 1024);
        dma16(rs_w1, dst, v1[0], soff);
        dma16(rs_w1, dst + 8 * 1024, v1[1], soff);
    };
    auto issue_w2 = [&](int hc, int h) {
        const unsigned soff = (unsigned)((h * 160 * FC_HID + hc * 64) * 2);
        const unsigned dst = lds0 + (unsigned)(FC_W2 + h * FC_S2);
#pragma unroll
        for (int j = 0; j < 3; ++j) dma16(rs_w2, dst + d2[j], v2[j], soff);
    };
    if (wave_s < 5) {
        const i32x4 rs_b1 = make_rsrc_words(p.b1, (unsigned)(2 * FC_HID * 2));
        dma16(rs_b1, lds0 + (unsigned)(FC_B1 + wave_s * 1024), (unsigned)(wave_s * 1024 + lane * 16), 0u);
    }
#pragma unroll
    for (int kt = 0; kt < FC_KT; ++kt) issue_w1(0, kt);

    {   // LayerNorm in registers: the two lanes l31 / l31 + 32 hold one row between them
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < FC_KT * 4; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) s += (float)xf[k][i];
        s += __shfl_xor(s, 32, 64);
        const float mean = s * (1.0f / FC_C);
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < FC_KT * 4; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float dl = (float)xf[k][i] - mean;
                q += dl * dl;
            }
        q += __shfl_xor(q, 32, 64);
        const float rstd = rsqrtf(q * (1.0f / FC_C) + p.eps);
        const float nmr = -mean * rstd;
#pragma unroll
        for (int k = 0; k < FC_KT * 4; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) xf[k][i] = (f16)fmaf((float)xf[k][i], rstd, nmr);
    }
    zero_acc2();

    int rd1[4], rd2[4], rdh[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        rd1[ks] = lds_off_kb<64>(wn * 64 + l31, ks * 2 + hi);
        rd2[ks] = FC_W2 + wn * FC_S2 + lds_off_kb<64>(l31, ks * 2 + hi);
        rdh[ks] = FC_HT + lds_off_kb<64>(wm * 32 + l31, ks * 2 + hi);
    }
    int wr_h[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) wr_h[g] = FC_HT + lds_off_kb<64>(wm * 32 + l31, wn * 4 + g) + 8 * hi;

    f32x16 acc1[2];
    auto stage1_tile = [&](int kt) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            U4H8 wv, wg;
            wv.u = *reinterpret_cast<const uint4*>(smem + kt * FC_S1 + rd1[ks]);
            wg.u = *reinterpret_cast<const uint4*>(smem + kt * FC_S1 + rd1[ks] + 32 * 128);
            acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv.h, xf[kt * 4 + ks], acc1[0], 0, 0, 0);
            acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wg.h, xf[kt * 4 + ks], acc1[1], 0, 0, 0);
        }
    };
    auto zero_acc1 = [&]() {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[t][r] = 0.f;
    };
    auto geglu_group = [&](int hc, int g, int hb) {
        const char* bp = smem + FC_B1 + (hc * 128 + wn * 64 + 4 * hi) * 2;
        U2H4 bv, bg, o;
        bv.u = *reinterpret_cast<const uint2*>(bp + 16 * g);
        bg.u = *reinterpret_cast<const uint2*>(bp + 16 * g + 64);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float v = acc1[0][g * 4 + q] + (float)bv.e[q];
            const float gt = acc1[1][g * 4 + q] + (float)bg.e[q];
            o.e[q] = (f16)(v * vd_gelu_erf(gt));
        }
        *reinterpret_cast<uint2*>(smem + wr_h[g] + hb * (FC_BM * 128)) = o.u;
    };
    auto stage2_step = [&](int ks, int hb) {
        U4H8 hf;
        hf.u = *reinterpret_cast<const uint4*>(smem + rdh[ks] + hb * (FC_BM * 128));
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            U4H8 wf;
            wf.u = *reinterpret_cast<const uint4*>(smem + rd2[ks] + j * 32 * 128);
            acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf.h, hf.h, acc2[j], 0, 0, 0);
        }
    };
    for (int hc = 0; hc < FC_NCH; ++hc) {
        const bool first = hc == 0, more = hc + 1 < FC_NCH;
        wait_vm<6>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (!first) {
            issue_w2(hc - 1, 0);
            issue_w2(hc - 1, 1);
        }
        zero_acc1();
        stage1_tile(0);
        stage1_tile(1);
        if (first) wait_vm<0>();
        else wait_vm<6>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (more) {
            issue_w1(hc + 1, 0);
            issue_w1(hc + 1, 1);
        }
        stage1_tile(2);
        stage1_tile(3);
        stage1_tile(4);
        if (!first) {
            if (more) wait_vm<4>();
            else wait_vm<0>();
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (more) {
            issue_w1(hc + 1, 2);
            issue_w1(hc + 1, 3);
            issue_w1(hc + 1, 4);
        }
        const int hb = hc & 1;
        if (!first) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                stage2_step(ks, hb ^ 1);
                geglu_group(hc, ks, hb);
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) geglu_group(hc, g, hb);
        }
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue_w2(FC_NCH - 1, 0);
    issue_w2(FC_NCH - 1, 1);
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) stage2_step(ks, (FC_NCH - 1) & 1);
    wait_vm<0>();
    __syncthreads();   // every wave is done with the slots: the tile re-uses that LDS

    // ================= epilogue: y = ff + b2 + x1 =================
    acc_to_tile(p.b2, 1.0f);
    __syncthreads();
    const f16* xres = PRE ? p.x1 : p.x;   // (PRE: written by this block's own threads above, long acknowledged)
    if constexpr (!POST) {
        segments(YES, NO, xres, p.y, p.nt_store != 0);
    } else {
        // ---- out = alpha (y Wp^T + bp) + r, + per-channel statistics of the stored rows
        segments(NO, YES, xres, nullptr, false);
        zero_acc2();
        wait_vm<0>();
        __syncthreads();
        const SegAdd rseg = seg_request(p.r);    // the residual of the last projection: in flight across its K loop
        proj_stage(p.wp);
        acc_to_tile(p.bp, p.alpha);
        wait_vm<0>();
        __syncthreads();
        const bool stats = p.out_stats != nullptr;
        seg_finish(YES, YES, rseg, p.y, p.nt_store != 0);
        if (stats) {
            __syncthreads();
            const int left = (p.M - m0) / FC_BM;
            emit_chan_stats<FC_C, FC_CS_LD, 512>(cs, reinterpret_cast<float*>(smem + FC_RING), tid, FC_BM, 1, left < 1 ? left : 1, p.out_stats,
                                                 (size_t)(m0 / FC_BM), FC_C, 0, p.stat_sums, p.img_rows);
        }
    }
}

template <bool PRE, bool POST>
int launch_ffc(const FCArgs& a, hipStream_t stream) {
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ff_chain_kernel<PRE, POST>), hipFuncAttributeMaxDynamicSharedMemorySize, FC_LDS);
        if (e != hipSuccess) {
            vd_set_error("vd_ff_chain_f16: cannot reserve %d bytes of LDS: %s", FC_LDS, hipGetErrorString(e));
            return VD_ERR_LAUNCH;
        }
        done.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL((ff_chain_kernel<PRE, POST>), dim3((unsigned)((a.M + FC_BM - 1) / FC_BM)), dim3(512), FC_LDS, stream, a);
    return vd_check_launch("vd_ff_chain_f16");
}

}  // namespace

extern "C" int vd_ff_chain_supported(int C) { return C == FC_C ? 1 : 0; }

extern "C" int vd_ff_chain_f16(const VdFfChain* c, hipStream_t stream) {
    VD_REQUIRE(c != nullptr, "vd_ff_chain_f16: null descriptor");
    VD_REQUIRE(c->C == FC_C, "vd_ff_chain_f16: inner width %d not instantiated (only %d)", c->C, FC_C);
    VD_REQUIRE(c->M > 0 && c->M < (1ll << 31) / FC_C, "vd_ff_chain_f16: bad row count %ld", (long)c->M);
    VD_REQUIRE(c->x && c->w1_packed && c->b1_packed && c->w2 && c->b2 && c->out, "vd_ff_chain_f16: null feed-forward operand");
    const bool pre = c->a != nullptr, post = c->wp != nullptr;
    VD_REQUIRE(pre || post, "vd_ff_chain_f16: neither projection given (use vd_ff_geglu_f16)");
    if (pre) VD_REQUIRE(c->wo && c->bo && c->x1_scratch, "vd_ff_chain_f16: the first projection needs wo, bo and the x1 scratch");
    if (post) VD_REQUIRE(c->bp && c->res, "vd_ff_chain_f16: the last projection needs bp and its residual");
    VD_REQUIRE(post || c->out_stats == nullptr, "vd_ff_chain_f16: out_stats describe the last projection's output");
    VD_REQUIRE(c->out_stats == nullptr || c->M % FC_BM == 0, "vd_ff_chain_f16: out_stats need M %% 128 == 0");
    VD_REQUIRE(c->stat_sums == nullptr || (c->out_stats != nullptr && ((size_t)c->stat_sums & 7) == 0 && c->stat_img_rows > 0 &&
                                           c->stat_img_rows % FC_BM == 0 && c->M % c->stat_img_rows == 0),
               "vd_ff_chain_f16: stat_sums rides on out_stats (whole images of stat_img_rows rows, a multiple of 128)");
    const size_t align16 = (size_t)c->x | (size_t)c->w1_packed | (size_t)c->w2 | (size_t)c->out | (size_t)c->a | (size_t)c->wo | (size_t)c->x1_scratch |
                           (size_t)c->wp | (size_t)c->res;
    const size_t align8 = (size_t)c->b1_packed | (size_t)c->b2 | (size_t)c->bo | (size_t)c->bp | (size_t)c->out_stats;
    VD_REQUIRE((align16 & 15) == 0 && (align8 & 7) == 0, "vd_ff_chain_f16: operands must be 16-byte aligned (biases / statistics 8)");
    FCArgs a;
    a.x = (const f16*)c->x; a.a = (const f16*)c->a; a.wo = (const f16*)c->wo; a.bo = (const f16*)c->bo; a.x1 = (f16*)c->x1_scratch;
    a.w1 = (const f16*)c->w1_packed; a.b1 = (const f16*)c->b1_packed; a.w2 = (const f16*)c->w2; a.b2 = (const f16*)c->b2;
    a.wp = (const f16*)c->wp; a.bp = (const f16*)c->bp; a.r = (const f16*)c->res; a.y = (f16*)c->out; a.out_stats = c->out_stats;
    a.stat_sums = reinterpret_cast<unsigned long long*>(c->stat_sums); a.img_rows = (int)c->stat_img_rows;
    a.M = (int)c->M; a.eps = c->ln_eps; a.alpha = c->alpha;
    a.nt_store = 1;   // non-temporal stores of write-once outputs
    if (pre && post) return launch_ffc<true, true>(a, stream);
    if (pre) return launch_ffc<true, false>(a, stream);
    return launch_ffc<false, true>(a, stream);
}
