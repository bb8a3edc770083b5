// vd_ff_geglu_f16: the whole gated feed-forward of a BasicTransformerBlock in ONE launch for inner width C = 320 (the 64x64
// level of the UNet, where the [M, 4C] GEGLU intermediate is 84 MB per layer):
//
//     y[m] = res[m] + ( v (*) gelu(g) ) W2^T + b2,      [v | g] = LayerNorm(x[m]) W1^T + b1
//
// Replaces, on the reference path, norm3 -> GEGLU.proj -> value * gelu(gate) -> net[2] Linear -> + x of
// /root/reference/lib/model_zoo/attention.py:37-64,214-218 (there: LayerNorm, Linear, chunk, F.gelu, mul, Linear, add as
// separate torch ops), and in this library the chain vd_row_stats_f16 -> vd_gemm_f16(GEGLU, LN fold) -> vd_gemm_f16(residual)
// whose [M, 4C] intermediate made a round trip through HBM.
//
// A block owns 128 rows.  What is resident where:
//   * registers: the block's rows of x as MFMA operand fragments (each wave: its 32 rows x 320 columns = 80 registers per
//     lane), layer-normalised IN PLACE once (mean / rstd from the fragments + one lane^32 exchange; gamma is folded into W1,
//     beta into b1 by the host), the [128 x 320] fp32 output accumulator (80 registers), the [32 x 64] GEGLU accumulators.
//   * LDS: seven weight-tile slots with a STATIC assignment -- slots 0..4 (16 KiB): the five 64-deep K tiles of the current
//     hidden chunk's W1 rows (128 packed rows = 64 hidden units as [32 value | 32 gate] pairs), slots 5 / 6 (20 KiB): the two
//     160-row halves of the chunk's W2 columns --, the [128 x 64] fp16 tile of v * gelu(g) that becomes the next operand,
//     and b1 (5 KiB: read with ds_read in the GEGLU epilogue, so no compiler-counted global load sits between the
//     hand-counted LDS-DMA requests).
//   The hidden dimension is walked in 20 chunks of 64; per chunk three barriers:
//     step A: K tiles 0, 1 of stage 1          (then refill slots 5, 6 with this chunk's W2 halves)
//     step B: K tiles 2, 3, 4 + GEGLU epilogue  (then refill slots 0, 1 with the NEXT chunk's W1 tiles)
//     step C: stage 2, out += h W2c^T           (then refill slots 2, 3, 4)
//   so every weight tile is requested two steps (~2500 cycles) before its first read, all waves issue the same number of
//   LDS-DMA pieces per refill (the 20 pieces of a W2 half go 3 per wave with 4 duplicates) and the waits are counted
//   (vmcnt 6 / 6 / 4): nothing in the loop waits for the youngest request.
// x is read once, y written once, weights stream from L2 (2.4 MB, shared by every block).
#include "gemm_kernel.h"

namespace {

constexpr int FF_C = 320;                 // model width this instantiation serves
constexpr int FF_KT = FF_C / 64;          // K tiles of stage 1
constexpr int FF_HID = 4 * FF_C;          // hidden units
constexpr int FF_NCH = FF_HID / 64;       // hidden chunks
constexpr int FF_BM = 128;
constexpr int FF_S1 = 16384;              // bytes of a W1 slot (128 rows x 128)
constexpr int FF_S2 = 20480;              // bytes of a W2 slot (160 rows x 128)
constexpr int FF_W2 = FF_KT * FF_S1;      // byte offset of the two W2 slots
constexpr int FF_HT = FF_W2 + 2 * FF_S2;  // ... of the h tile [128][64]
constexpr int FF_B1 = FF_HT + 2 * FF_BM * 128;   // ... of b1 (all 8C packed entries, fp16); two h tiles (double-buffered)
constexpr int FF_LDS_MAIN = FF_B1 + 2 * FF_HID * 2;
constexpr int FF_CS_LD = FF_C + 8;
constexpr int FF_LDS_EPI = FF_BM * FF_CS_LD * 2;
constexpr int FF_LDS = FF_LDS_MAIN > FF_LDS_EPI ? FF_LDS_MAIN : FF_LDS_EPI;
static_assert(FF_LDS <= 160 * 1024, "LDS budget");

struct FFArgs {
    const f16* x;      // [M][C]
    const f16* w1;     // [8C][C]  gamma-folded, GEGLU-packed ([32 value | 32 gate] per 64 rows)
    const f16* b1;     // [8C]     beta-folded, packed likewise
    const f16* w2;     // [C][4C]
    const f16* b2;     // [C]
    const f16* res;    // [M][C]
    f16* y;            // [M][C]
    int M;
    float eps;
    int nt_store;
};

// VER 0: stage 2 of a chunk right behind its GEGLU epilogue (both waves of a SIMD sit in the VALU-only epilogue at the same
// time).  VER 1: the epilogue of chunk n shares step C with stage 2 of chunk n - 1 (h tile double-buffered), so its ~350
// VALU instructions per wave issue in the shadow of 20 MFMAs instead of in front of them.
// ABL > 0: timing-only ablations (wrong results; VD_FF_ABL): 1 no LDS-DMA in the loop, 2 no GELU arithmetic, 3 no barriers
// in the loop, 4 no stage-2 MFMAs, 5 no stage-1 MFMAs.
template <int VER, int ABL = 0>
__global__ __launch_bounds__(512, 2) void ff_geglu_kernel(const FFArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave >> 1, wn = wave & 1;
    const int hi = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.x * FF_BM;

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const i32x4 rs_w1 = make_rsrc_words(p.w1, (unsigned)(2 * FF_HID * FF_C * 2));
    const i32x4 rs_w2 = make_rsrc_words(p.w2, (unsigned)(FF_C * FF_HID * 2));

    // ---- DMA sources.  W1 tile: 128 rows = 16 pieces of 8 rows, wave w issues pieces w and w + 8.  W2 half: 160 rows =
    // 20 pieces, wave w issues w, w + 8 and (w + 16 < 20 ? w + 16 : w + 8 again: identical bytes to the identical place).
    unsigned v1[2], v2[3], d2[3];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = (j * 8 + wave) * 8 + (lane >> 3);
        v1[j] = (unsigned)((r * FF_C + (((lane & 7) ^ ((r >> 1) & 7)) << 3)) * 2);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        int q = j * 8 + wave;
        if (q >= 20) q -= 8;
        const int r = q * 8 + (lane >> 3);
        v2[j] = (unsigned)((r * FF_HID + (((lane & 7) ^ ((r >> 1) & 7)) << 3)) * 2);
        d2[j] = (unsigned)(__builtin_amdgcn_readfirstlane(q) * 1024);
    }
    auto issue_w1 = [&](int hc, int kt) {   // K tile kt of chunk hc -> slot kt
        if (ABL == 1 && hc > 0) return;
        const unsigned soff = (unsigned)((hc * 128 * FF_C + kt * 64) * 2);
        const unsigned dst = lds0 + (unsigned)(kt * FF_S1 + wave_s * 1024);
        dma16(rs_w1, dst, v1[0], soff);
        dma16(rs_w1, dst + 8 * 1024, v1[1], soff);
    };
    auto issue_w2 = [&](int hc, int h) {    // rows 160 h .. 160 h + 159 of W2, columns of chunk hc -> slot 5 + h
        if (ABL == 1 && hc > 0) return;
        const unsigned soff = (unsigned)((h * 160 * FF_HID + hc * 64) * 2);
        const unsigned dst = lds0 + (unsigned)(FF_W2 + h * FF_S2);
#pragma unroll
        for (int j = 0; j < 3; ++j) dma16(rs_w2, dst + d2[j], v2[j], soff);
    };

    // ---- x fragments: xf[kt * 4 + ks] = x[row][kt * 64 + ks * 16 + hi * 8 .. + 8] (B operand of stage 1)
    const int row = m0 + wm * 32 + l31;
    const int rowc = row < p.M ? row : p.M - 1;
    f16x8 xf[FF_KT * 4];
    {
        const f16* xr = p.x + (size_t)rowc * FF_C + hi * 8;
#pragma unroll
        for (int k = 0; k < FF_KT * 4; ++k) {
            U4H8 t;
            t.u = *reinterpret_cast<const uint4*>(xr + k * 16);
            xf[k] = t.h;
        }
    }
    // b1 -> LDS: 5 KiB = 5 pieces, one each for waves 0..4 (older than every counted request below, published by the
    // first barrier of the loop)
    if (wave_s < 5) {
        const i32x4 rs_b1 = make_rsrc_words(p.b1, (unsigned)(2 * FF_HID * 2));
        dma16(rs_b1, lds0 + (unsigned)(FF_B1 + wave_s * 1024), (unsigned)(wave_s * 1024 + lane * 16), 0u);
    }
    // first chunk's W1 tiles, in the steady-state order (0, 1 | 2, 3, 4)
#pragma unroll
    for (int kt = 0; kt < FF_KT; ++kt) issue_w1(0, kt);

    // ---- LayerNorm in registers: the two lanes l31 / l31 + 32 hold one row between them
    {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < FF_KT * 4; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) s += (float)xf[k][i];
        s += __shfl_xor(s, 32, 64);
        const float mean = s * (1.0f / FF_C);
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < FF_KT * 4; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float dl = (float)xf[k][i] - mean;
                q += dl * dl;
            }
        q += __shfl_xor(q, 32, 64);
        const float rstd = rsqrtf(q * (1.0f / FF_C) + p.eps);
        const float nmr = -mean * rstd;
#pragma unroll
        for (int k = 0; k < FF_KT * 4; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) xf[k][i] = (f16)fmaf((float)xf[k][i], rstd, nmr);
    }

    f32x16 acc2[5];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;

    // LDS read offsets
    int rd1[4], rd2[4], rdh[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        rd1[ks] = lds_off_kb<64>(wn * 64 + l31, ks * 2 + hi);                       // + kt * FF_S1 + t * 32 * 128
        rd2[ks] = FF_W2 + wn * FF_S2 + lds_off_kb<64>(l31, ks * 2 + hi);            // + j * 32 * 128
        rdh[ks] = FF_HT + lds_off_kb<64>(wm * 32 + l31, ks * 2 + hi);
    }
    int wr_h[4];   // h tile: 4 consecutive columns (8 bytes) at column wn * 32 + 8 g + 4 hi of row wm * 32 + l31
#pragma unroll
    for (int g = 0; g < 4; ++g) wr_h[g] = FF_HT + lds_off_kb<64>(wm * 32 + l31, wn * 4 + g) + 8 * hi;

    f32x16 acc1[2];
    auto stage1_tile = [&](int kt) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            U4H8 wv, wg;
            wv.u = *reinterpret_cast<const uint4*>(smem + kt * FF_S1 + rd1[ks]);
            wg.u = *reinterpret_cast<const uint4*>(smem + kt * FF_S1 + rd1[ks] + 32 * 128);
            if constexpr (ABL == 5) {
                acc1[0][ks] += (float)wv.h[0] * (float)xf[kt * 4 + ks][0];
                acc1[1][ks] += (float)wg.h[0] * (float)xf[kt * 4 + ks][1];
            } else {
                acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv.h, xf[kt * 4 + ks], acc1[0], 0, 0, 0);
                acc1[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wg.h, xf[kt * 4 + ks], acc1[1], 0, 0, 0);
            }
        }
    };

    auto zero_acc1 = [&]() {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[t][r] = 0.f;
    };
    // GEGLU epilogue, group g of chunk hc: 4 consecutive hidden units of this lane's row -> h tile `hb`
    auto geglu_group = [&](int hc, int g, int hb) {
        const char* bp = smem + FF_B1 + (hc * 128 + wn * 64 + 4 * hi) * 2;
        U2H4 bv, bg, o;
        bv.u = *reinterpret_cast<const uint2*>(bp + 16 * g);
        bg.u = *reinterpret_cast<const uint2*>(bp + 16 * g + 64);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float v = acc1[0][g * 4 + q] + (float)bv.e[q];
            const float gt = acc1[1][g * 4 + q] + (float)bg.e[q];
            o.e[q] = ABL == 2 ? (f16)(v + gt) : (f16)(v * vd_gelu_erf(gt));
        }
        *reinterpret_cast<uint2*>(smem + wr_h[g] + hb * (FF_BM * 128)) = o.u;
    };
    auto stage2_step = [&](int ks, int hb) {
        U4H8 hf;
        hf.u = *reinterpret_cast<const uint4*>(smem + rdh[ks] + hb * (FF_BM * 128));
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            U4H8 wf;
            wf.u = *reinterpret_cast<const uint4*>(smem + rd2[ks] + j * 32 * 128);
            if constexpr (ABL == 4) acc2[j][ks] += (float)wf.h[0] * (float)hf.h[j];
            else acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf.h, hf.h, acc2[j], 0, 0, 0);
        }
    };
    auto loop_barrier = [&]() { if constexpr (ABL != 3) __builtin_amdgcn_s_barrier(); };

    if constexpr (VER == 0) {
        for (int hc = 0; hc < FF_NCH; ++hc) {
            const bool more = hc + 1 < FF_NCH;
            // ---- step A: K tiles 0, 1 (requested two steps ago; the three tiles requested after them may still be in flight)
            wait_vm<6>();
            loop_barrier();
            asm volatile("" ::: "memory");
            issue_w2(hc, 0);
            issue_w2(hc, 1);
            zero_acc1();
            stage1_tile(0);
            stage1_tile(1);
            // ---- step B: K tiles 2, 3, 4, then value * gelu(gate) -> h tile
            wait_vm<6>();
            loop_barrier();
            asm volatile("" ::: "memory");
            if (more) {
                issue_w1(hc + 1, 0);
                issue_w1(hc + 1, 1);
            }
            stage1_tile(2);
            stage1_tile(3);
            stage1_tile(4);
#pragma unroll
            for (int g = 0; g < 4; ++g) geglu_group(hc, g, 0);
            // ---- step C: out += h W2c^T
            if (more) wait_vm<4>();
            else wait_vm<0>();
            loop_barrier();
            asm volatile("" ::: "memory");
            if (more) {
                issue_w1(hc + 1, 2);
                issue_w1(hc + 1, 3);
                issue_w1(hc + 1, 4);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) stage2_step(ks, 0);
        }
    } else {
        // W2 halves of chunk n are requested at step A of iteration n + 1 and read at its step C (stage 2 runs one chunk late)
        for (int hc = 0; hc < FF_NCH; ++hc) {
            const bool first = hc == 0, more = hc + 1 < FF_NCH;
            // ---- step A: K tiles 0, 1
            wait_vm<6>();
            loop_barrier();
            asm volatile("" ::: "memory");
            if (!first) {
                issue_w2(hc - 1, 0);
                issue_w2(hc - 1, 1);
            }
            zero_acc1();
            stage1_tile(0);
            stage1_tile(1);
            // ---- step B: K tiles 2, 3, 4
            if (first) wait_vm<0>();
            else wait_vm<6>();
            loop_barrier();
            asm volatile("" ::: "memory");
            if (more) {
                issue_w1(hc + 1, 0);
                issue_w1(hc + 1, 1);
            }
            stage1_tile(2);
            stage1_tile(3);
            stage1_tile(4);
            // ---- step C: stage 2 of the PREVIOUS chunk with this chunk's GEGLU epilogue in its shadow
            if (!first) {
                if (more) wait_vm<4>();
                else wait_vm<0>();
            }
            loop_barrier();
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
        // ---- tail: stage 2 of the last chunk
        __builtin_amdgcn_s_barrier();   // every wave has left step C: the W2 slots are free, the last h tile is written
        asm volatile("" ::: "memory");
        issue_w2(FF_NCH - 1, 0);
        issue_w2(FF_NCH - 1, 1);
        wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) stage2_step(ks, (FF_NCH - 1) & 1);
    }
    wait_vm<0>();
    __syncthreads();   // every wave is done with the slots: the output tile re-uses that LDS

    // ---- epilogue: + b2 -> fp16 tile in LDS -> 16-byte row segments + residual -> y
    f16* cs = reinterpret_cast<f16*>(smem);
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int col = wn * 160 + j * 32 + 8 * g + 4 * hi;
            U2H4 b, o;
            b.u = *reinterpret_cast<const uint2*>(p.b2 + col);
#pragma unroll
            for (int q = 0; q < 4; ++q) o.e[q] = (f16)(acc2[j][g * 4 + q] + (float)b.e[q]);
            *reinterpret_cast<uint2*>(cs + (wm * 32 + l31) * FF_CS_LD + col) = o.u;
        }
    __syncthreads();
    constexpr int CH = FF_C / 8;                       // 40 segments per row
    constexpr int PER = FF_BM * CH / 512;              // 10 per thread
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int sgm = tid + k * 512;
        const int r = sgm / CH, cc = (sgm % CH) * 8;
        const int grow = m0 + r;
        if (grow < p.M) {
            U4H8 t, a, o;
            t.u = *reinterpret_cast<const uint4*>(cs + r * FF_CS_LD + cc);
            a.u = *reinterpret_cast<const uint4*>(p.res + (size_t)grow * FF_C + cc);
#pragma unroll
            for (int q = 0; q < 8; ++q) o.e[q] = (f16)((float)t.e[q] + (float)a.e[q]);
            f16* dst = p.y + (size_t)grow * FF_C + cc;
            if (p.nt_store) vd_store16_nt(dst, o.u);
            else *reinterpret_cast<uint4*>(dst) = o.u;
        }
    }
}

}  // namespace

extern "C" int vd_ff_geglu_supported(int C) { return C == FF_C ? 1 : 0; }

extern "C" int vd_ff_geglu_f16(const void* x, const void* w1_packed, const void* b1_packed, const void* w2, const void* b2,
                               const void* res, void* y, int64_t M, int C, float ln_eps, hipStream_t stream) {
    VD_REQUIRE(x && w1_packed && b1_packed && w2 && b2 && res && y, "vd_ff_geglu_f16: null pointer");
    VD_REQUIRE(C == FF_C, "vd_ff_geglu_f16: inner width %d not instantiated (only %d; use the vd_gemm_f16 chain)", C, FF_C);
    VD_REQUIRE(M > 0 && M < (1ll << 31) / FF_C, "vd_ff_geglu_f16: bad row count %ld", (long)M);
    VD_REQUIRE((((size_t)x | (size_t)w1_packed | (size_t)w2 | (size_t)res | (size_t)y) & 15) == 0 && (((size_t)b1_packed | (size_t)b2) & 7) == 0,
               "vd_ff_geglu_f16: operands must be 16-byte aligned (biases 8)");
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ff_geglu_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ff_geglu_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS);
#ifdef VD_FF_ABLATIONS
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ff_geglu_kernel<0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ff_geglu_kernel<0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ff_geglu_kernel<0, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ff_geglu_kernel<0, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ff_geglu_kernel<0, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS);
#endif
        if (e != hipSuccess) {
            vd_set_error("vd_ff_geglu_f16: cannot reserve %d bytes of LDS: %s", FF_LDS, hipGetErrorString(e));
            return VD_ERR_LAUNCH;
        }
        done.fetch_or(bit, std::memory_order_release);
    }
    FFArgs a;
    a.x = (const f16*)x; a.w1 = (const f16*)w1_packed; a.b1 = (const f16*)b1_packed; a.w2 = (const f16*)w2; a.b2 = (const f16*)b2;
    a.res = (const f16*)res; a.y = (f16*)y; a.M = (int)M; a.eps = ln_eps;
    a.nt_store = 1;   // non-temporal stores of write-once outputs
    const dim3 grid((unsigned)((M + FF_BM - 1) / FF_BM));
#ifdef VD_FF_ABLATIONS
    static const char* abl_env = getenv("VD_FF_ABL");
    const int abl = abl_env ? atoi(abl_env) : 0;
    if (abl == 1) { hipLaunchKernelGGL((ff_geglu_kernel<0, 1>), grid, dim3(512), FF_LDS, stream, a); return vd_check_launch("ff abl"); }
    if (abl == 2) { hipLaunchKernelGGL((ff_geglu_kernel<0, 2>), grid, dim3(512), FF_LDS, stream, a); return vd_check_launch("ff abl"); }
    if (abl == 3) { hipLaunchKernelGGL((ff_geglu_kernel<0, 3>), grid, dim3(512), FF_LDS, stream, a); return vd_check_launch("ff abl"); }
    if (abl == 4) { hipLaunchKernelGGL((ff_geglu_kernel<0, 4>), grid, dim3(512), FF_LDS, stream, a); return vd_check_launch("ff abl"); }
    if (abl == 5) { hipLaunchKernelGGL((ff_geglu_kernel<0, 5>), grid, dim3(512), FF_LDS, stream, a); return vd_check_launch("ff abl"); }
#endif
    hipLaunchKernelGGL(ff_geglu_kernel<1>, grid, dim3(512), FF_LDS, stream, a);
    return vd_check_launch("vd_ff_geglu_f16");
}
