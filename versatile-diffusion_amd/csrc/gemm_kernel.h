// fp16 MFMA GEMM / implicit-GEMM convolution kernel template for gfx950 (CDNA4).  Included by gemm.hip (planner,
// 2-waves-per-SIMD instances) and gemm_big.hip (one-wave-per-SIMD instances with large per-wave tiles).
//
//   out[m][n] = epilogue( sum_k A[m][k] * W[n][k] )
//
// A is never materialised for convolutions: the tile loader gathers the (kh,kw) tap of an NHWC activation (optionally
// the channel-concatenation of two tensors, optionally nearest-2x upsampled, optionally strided) straight into LDS.
// W is K-contiguous ([N][K], K ordered (kh, kw, cin)): torch Linear weights as they are, conv weights repacked once.
//
// Replaces on the reference path (all stock torch ops there):
//   nn.Conv2d 3x3 / 1x1 in ResBlock, Up/Downsample, SpatialTransformer.proj_in/out
//     (/root/reference/lib/model_zoo/openaimodel.py:89-117,133-159,254-274, attention.py:255-266)
//   nn.Linear in CrossAttention / GEGLU FeedForward / time_embed / emb_layers (attention.py:37-64,170-193,
//     openaimodel.py:2627-2633), nn.LayerNorm in front of them (attention.py:205-218) through VD_EPI_LNFOLD
//   the VAE convs and AttnBlock bmm's (autokl_modules.py:82-202)
//
// Structure of a block (BM x BN output tile, NT/64 waves, wave tile WM x WN = MI x NI MFMA tiles of 32x32):
//   * K is walked in KB-deep tiles through a ring of STAGES LDS buffers.  Tiles travel global -> LDS by LDS-DMA
//     (buffer_load ... lds: no VGPR round trip, no ds_write); the XOR swizzle of the LDS image is applied on the source
//     side.  Default main loop ("burst"): wait for tile i, barrier, issue ALL DMA instructions of tile i + STAGES - 1,
//     then the fragment reads and MFMAs of tile i.  -DVD_GEMM_PIPELINED builds the software-pipelined alternative (DMA
//     issued in 1-KiB pieces between MFMA groups, operand fragments double-buffered across the per-tile barrier): faster
//     in a warm sweep, 4 % slower inside the UNet forward where weights stream cold from HBM.
//   * v_mfma_f32_32x32x16_f16, fp32 accumulation, operands swapped (MFMA A operand = W rows, B operand = activation
//     rows) so a lane owns ONE output row and 4 consecutive columns per register group: bias / LayerNorm fold /
//     activation / GEGLU gating / alpha run in registers, the tile is staged through LDS and leaves as 16-byte row
//     segments with the per-batch row vector and the residual added on the way out.
//   * Larger wave tiles cut LDS traffic per MFMA (reads: (MI+NI)/(MI*NI) KiB per MFMA; DMA writes: (BM+BN)*KB*2 bytes per
//     tile): the one-wave-per-SIMD and 64x128 / 64x160-per-wave instances of gemm_big.hip exist for that and for the
//     experiments DESIGN.md reports; the UNet's shapes do not reward them.
//   * LayerNorm fold (LNF instances): (mean, rstd) per row from d.ln_stats, the block's colsum slice from LDS.
//   * split-K: fp32 slabs + splitk_reduce_kernel (gemm.hip); optional in-kernel reduction by the last-arriving block
//     (d.sync, agent-scope atomics) -- correct, slower, off by default.
#pragma once
#include "vd_common.h"
#include "../../include/vd_hip.h"
#include <atomic>
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int BK = 64;  // the planner's K unit (halfs); kernels walk K in KB = 64 or 32

struct GemmArgs {
    VdGemmDesc d;
    int tiles_m, tiles_n, kt_total, kt_per_split;
    unsigned a0_bytes, a1_bytes, w_bytes;  // per-batch operand extents for the buffer descriptors (< 2^31)
    int plain;                             // 1x1, stride 1, no pad / upsample, output grid == input grid
    int nt_store;                          // non-temporal output stores (streaming results that nobody re-reads soon)
    int mfast;                             // XCD tile runs walk m fastest (tiles of one weight column panel share an L2)
    int stat_rows;                         // d.out_stats: rows per statistics partial (0 = none emitted by this launch)
    int xcd_local;                         // halo conv, ticketed split: the blocks of a tile share an XCD (L2-scope exchange)
    int hoist;                             // gemm_f16_kernel, small tiles: epilogue operands requested in front of the main loop
#ifdef VD_TIMELINE
    unsigned long long* tl;                // development build (tools/probes/gemm_timeline.py): 8 stamps per block, or null
#endif
};

// Development builds with -DVD_TIMELINE (VD_EXTRA_DEFS, VD_BUILD_OUT): wall-clock stamps (s_memrealtime, 100 MHz) at the phase
// boundaries of a block, kept in scalar registers and stored by thread 0 at the very end -- no memory operation is added
// inside the hand-counted vmcnt regions.  The product library carries none of it.
#ifdef VD_TIMELINE
#define VD_TL_DECL unsigned long long vd_tl_t[6] = {0, 0, 0, 0, 0, 0}
#define VD_TL(i) vd_tl_t[i] = wall_clock64()
#define VD_TL_FLUSH(ptr)                                                                                              \
    do {                                                                                                              \
        if ((ptr) != nullptr && threadIdx.x == 0) {                                                                   \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                          \
            unsigned long long* o_ = (ptr) + ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8; \
            for (int i_ = 0; i_ < 6; ++i_) o_[i_] = vd_tl_t[i_];                                                      \
            o_[6] = wall_clock64();                                                                                   \
            o_[7] = (unsigned long long)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15);                             \
        }                                                                                                             \
    } while (0)
#else
#define VD_TL_DECL do {} while (0)
#define VD_TL(i) do {} while (0)
#define VD_TL_FLUSH(ptr) do {} while (0)
#endif

constexpr unsigned OOB_OFFSET = 0x80000000u;  // beyond every descriptor's num_records -> hardware returns zeros

__device__ __forceinline__ void vd_store16_nt(void* p, uint4 v) {
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<u32x4_t*>(p));
}

// swizzled byte offset of (row r, 16-byte slot s) inside a [rows][KB] f16 LDS tile.
// KB = 64: 128-byte rows, 8 slots, key (r >> 1) & 7: the 16 rows of a ds_read_b128 lane group cover both halves of the
//          256-byte bank row (r & 1) x 8 distinct keys.
// KB = 32: 64-byte rows, 4 slots, key (r >> 2) & 3: rows 4a..4a+3 fill one bank row, a & 3 spreads the slot.
template <int KB>
__device__ __forceinline__ int lds_swz(int r) { return KB == 64 ? ((r >> 1) & 7) : ((r >> 2) & 3); }
template <int KB>
__device__ __forceinline__ int lds_off_kb(int r, int s) { return r * (KB * 2) + ((s ^ lds_swz<KB>(r)) << 4); }

struct EpiCtx {
    const f16* bias;
    const f16* rowvec;
    const f16* res;
    void* out;
    unsigned long long* row_sums;   // VdGemmDesc.row_sums (gemm_f16_kernel's part 2 only), else null
    int N, ldc, ldr, rows_per_batch, flags, act;
    float alpha;
};

__device__ __forceinline__ EpiCtx make_epi(const VdGemmDesc& d, int z) {
    EpiCtx e;
    e.bias = reinterpret_cast<const f16*>(d.bias);
    e.rowvec = reinterpret_cast<const f16*>(d.rowvec);
    e.res = reinterpret_cast<const f16*>(d.res) + (size_t)z * d.stride_res;
    if (d.flags & VD_EPI_OUT_F32)
        e.out = reinterpret_cast<float*>(d.out) + (size_t)z * d.stride_out;
    else
        e.out = reinterpret_cast<f16*>(d.out) + (size_t)z * d.stride_out;
    e.row_sums = d.row_sums == nullptr ? nullptr : reinterpret_cast<unsigned long long*>(d.row_sums) + (size_t)z * d.M * 2;
    e.N = (d.act == VD_ACT_GEGLU) ? d.N / 2 : d.N;
    e.ldc = d.ldc;
    e.ldr = d.ldr;
    e.rows_per_batch = d.rows_per_batch > 0 ? d.rows_per_batch : 1;
    e.flags = d.flags;
    e.act = d.act;
    e.alpha = d.alpha;
    return e;
}

// bias of 4 consecutive output columns (zeros where there is none / past N)
__device__ __forceinline__ U2H4 epi_load_bias4(const EpiCtx& e, int N, int col) {
    U2H4 t;
    t.u = make_uint2(0, 0);
    if ((e.flags & VD_EPI_BIAS) && !(e.flags & VD_EPI_BIAS_ALONG_M)) {
        if (col + 4 <= N && (N & 3) == 0) {
            t.u = *reinterpret_cast<const uint2*>(e.bias + col);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (col + q < N) t.e[q] = e.bias[col + q];
        }
    }
    return t;
}

// The three fused activations are all x * sigmoid(s(x)) with s(x) = x (c1 + c3 x^2): quick GELU c1 = 1.702; SiLU c1 = 1; tanh
// GELU 0.5 x (1 + tanh(u)) = x sigmoid(2 u), c1 = 2 * 0.79788456, c3 = c1 * 0.044715.  One compact body (a multiply-add, v_exp,
// v_rcp) with the wave-uniform constants in scalar registers, instead of three inlined bodies with IEEE divisions: the
// epilogues unroll this 16-80 times per lane and the code of the activations NOT taken was what the instruction fetch of
// the ACT_NONE path hopped over (conv_halo_kernel.h, part 1).
__device__ __forceinline__ float apply_act(int act, float v) {
    if (act == VD_ACT_NONE) return v;
    const float c1 = act == VD_ACT_QUICK_GELU ? 1.702f : (act == VD_ACT_SILU ? 1.0f : 1.5957691216057308f);
    const float c3 = act == VD_ACT_GELU_TANH ? 0.07135481627260025f : 0.0f;
    const float sarg = v * fmaf(c3, v * v, c1);
    return v * __builtin_amdgcn_rcpf(1.0f + __expf(-sarg));
}

// Second half of the epilogue for 8 consecutive output columns of one row (values already carry
// bias / activation / alpha): + rowvec[batch] (+ residual) and the 16-byte store.
__device__ __forceinline__ void epi_finish8(const EpiCtx& e, int row, int col, float* v) {
    const bool full = (col + 8 <= e.N) && ((e.N & 7) == 0);
    if (e.flags & VD_EPI_ROWVEC) {
        const f16* rv = e.rowvec + (size_t)(row / e.rows_per_batch) * e.N + col;
        if (full) {
            U4H8 t;
            t.u = *reinterpret_cast<const uint4*>(rv);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += (float)t.e[i];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (col + i < e.N) v[i] += (float)rv[i];
        }
    }
    if (e.flags & VD_EPI_RESIDUAL) {
        const f16* rp = e.res + (size_t)row * e.ldr + col;
        if (full && ((e.ldr & 7) == 0)) {
            U4H8 t;
            t.u = *reinterpret_cast<const uint4*>(rp);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] += (float)t.e[i];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (col + i < e.N) v[i] += (float)rp[i];
        }
    }
    f16* op = reinterpret_cast<f16*>(e.out) + (size_t)row * e.ldc + col;
    if (full && ((e.ldc & 7) == 0)) {
        U4H8 t;
#pragma unroll
        for (int i = 0; i < 8; ++i) t.e[i] = (f16)v[i];
        *reinterpret_cast<uint4*>(op) = t.u;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (col + i < e.N) op[i] = (f16)v[i];
    }
}

// Full fp32 epilogue for 8 columns (split-K reduce kernel): bias, rowvec, act, alpha, residual, store.
__device__ __forceinline__ void epi_store8(const EpiCtx& e, int row, int col, float* v) {
    if (e.flags & VD_EPI_BIAS) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (col + i < e.N) v[i] += (float)((e.flags & VD_EPI_BIAS_ALONG_M) ? e.bias[row] : e.bias[col + i]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = apply_act(e.act, v[i]) * e.alpha;
    if (e.flags & VD_EPI_OUT_F32) {
        float* op = reinterpret_cast<float*>(e.out) + (size_t)row * e.ldc + col;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (col + i < e.N) op[i] = v[i];
        return;
    }
    epi_finish8(e, row, col, v);
}

// epi_store8 for launches whose 8-column groups are whole and vector-aligned (the split-K reduce kernels): the operands are
// REQUESTED first, branch-free (epi8_request: three 16-byte buffer loads, a disabled operand has an empty descriptor), and
// consumed later (epi8_finish), so they overlap the slab reads.  epi_store8 itself reads the bias element by element behind
// conditions -- hipcc makes that 8 serial round trips per call, ~5 of the 7.7 us of a reduce launch.
struct Epi8Ops { uint4 bias, rv, res; };
__device__ __forceinline__ bool epi8_fast(const EpiCtx& e) {
    return (e.N & 7) == 0 && (e.ldr & 7) == 0 && (e.ldc & 7) == 0 && !(e.flags & (VD_EPI_OUT_F32 | VD_EPI_BIAS_ALONG_M));
}
// (enable = false: empty descriptors, zeros -- NOT a branch around the requests: hipcc would wait for each one behind it)
__device__ __forceinline__ void epi8_request(const EpiCtx& e, int row, int col, Epi8Ops& o, bool enable) {
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(e.bias), 0, (enable && (e.flags & VD_EPI_BIAS)) ? e.N * 2 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(e.rowvec), 0, (enable && (e.flags & VD_EPI_ROWVEC)) ? 0x7fffffff : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(e.res), 0, (enable && (e.flags & VD_EPI_RESIDUAL)) ? 0x7fffffff : 0, 0x00020000);
    const u32x4_t b = __builtin_amdgcn_raw_buffer_load_b128(rs_b, col * 2, 0, 0);
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs_v, ((row / e.rows_per_batch) * e.N + col) * 2, 0, 0);
    const u32x4_t r = __builtin_amdgcn_raw_buffer_load_b128(rs_r, (row * e.ldr + col) * 2, 0, 0);
    o.bias = make_uint4(b[0], b[1], b[2], b[3]);
    o.rv = make_uint4(v[0], v[1], v[2], v[3]);
    o.res = make_uint4(r[0], r[1], r[2], r[3]);
}
// v: the reduced accumulators in, the values before the fp16 conversion out (as epi_store8 leaves them)
__device__ __forceinline__ void epi8_finish(const EpiCtx& e, int row, int col, float* v, const Epi8Ops& o) {
    U4H8 b, rv, rs, out;
    b.u = o.bias; rv.u = o.rv; rs.u = o.res;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = apply_act(e.act, v[i] + (float)b.e[i]) * e.alpha;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += (float)rv.e[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += (float)rs.e[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) out.e[i] = (f16)v[i];
    *reinterpret_cast<uint4*>(reinterpret_cast<f16*>(e.out) + (size_t)row * e.ldc + col) = out.u;
}

// ---- per-channel statistics of the stored tile for a consuming GroupNorm (VdGemmDesc.out_stats, csrc/gn_fused.hip) ----
// The fp16 tile cs[rows][CS_LD] holds the FINAL values (part 2 of the epilogue writes what it stores back into the tile).
// For each of `nsub` blocks of R consecutive tile rows and each of the BN columns: (mean, M2 = sum (x - mean)^2) over the R
// rows, accumulated as shifted sums around the block's first row (no cancellation for |mean| >> sigma), written to
// out[(pbase + s) * N + n0 + c] for s < pvalid.  The first 256 threads work: BN / 8 column octets x STAT_LANES row lanes, the
// lanes meet in `red` (fp32 [STAT_LANES][BN][2], behind the tile).
constexpr int stat_lanes(int bn) { return 256 / (bn / 8) < 32 ? 256 / (bn / 8) : 32; }
constexpr int stat_lds_bytes(int bm, int bn) { return bm * (bn + 8) * 2 + stat_lanes(bn) * bn * 8; }

// VdGemmDesc.stat_sums (ABI 7): the partial (mean, M2 over R rows) of channel n of image img as fixed-point (sum, sum of squares)
// added to sums[img][n]; the products are formed in fp64 so that the shifted partial loses nothing on the way
constexpr double GN_SUM_SCALE = 4294967296.0, GN_SQ_SCALE = 65536.0;   // 2^32, 2^16: |sum| < 2^30, sum of squares < 2^46 per image
__device__ __forceinline__ void gn_sums_add(unsigned long long* sums, size_t img, int N, int n, float mean, float m2, int R) {
    const double m = (double)mean, r = (double)R;
    unsigned long long* p = sums + (img * (size_t)N + n) * 2;
    atomicAdd(p, (unsigned long long)__double2ll_rn(r * m * GN_SUM_SCALE));
    atomicAdd(p + 1, (unsigned long long)__double2ll_rn((fma(r * m, m, (double)m2)) * GN_SQ_SCALE));
}

template <int BN, int CS_LD, int NT>
__device__ __forceinline__ void emit_chan_stats(const f16* cs, float* red, int tid, int R, int nsub, int pvalid, float* out,
                                                size_t pbase, int N, int n0, unsigned long long* sums = nullptr, int img_rows = 0) {
    constexpr int OCT = BN / 8;
    constexpr int LANES = stat_lanes(BN);
    const int co = tid % OCT, rl = tid / OCT;
    for (int s = 0; s < nsub; ++s) {
        if (s > 0) __syncthreads();   // `red` is re-used
        const f16* base = cs + (size_t)s * R * CS_LD;
        if (rl < LANES) {
            U4H8 kk;
            kk.u = *reinterpret_cast<const uint4*>(base + co * 8);
            float S[8], Q[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) S[q] = Q[q] = 0.f;
            for (int r = rl; r < R; r += LANES) {
                U4H8 t;
                t.u = *reinterpret_cast<const uint4*>(base + r * CS_LD + co * 8);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float v = (float)t.e[q] - (float)kk.e[q];
                    S[q] += v;
                    Q[q] = fmaf(v, v, Q[q]);
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) *reinterpret_cast<float2*>(red + (rl * BN + co * 8 + q) * 2) = make_float2(S[q], Q[q]);
        }
        __syncthreads();
        if (s < pvalid) {
            for (int c = tid; c < BN; c += NT) {
                if (n0 + c < N) {
                    float S = 0.f, Q = 0.f;
#pragma unroll 4
                    for (int l = 0; l < LANES; ++l) {
                        const float2 v = *reinterpret_cast<const float2*>(red + (l * BN + c) * 2);
                        S += v.x;
                        Q += v.y;
                    }
                    const float n = (float)R;
                    const float mean = (float)base[c] + S / n, m2 = fmaxf(Q - S * S / n, 0.f);
                    *reinterpret_cast<float2*>(out + ((pbase + s) * (size_t)N + n0 + c) * 2) = make_float2(mean, m2);
                    if (sums != nullptr) gn_sums_add(sums, ((pbase + s) * (size_t)R) / (size_t)img_rows, N, n0 + c, mean, m2, R);
                }
            }
        }
    }
}

// Epilogue part 2: the block's fp16 tile (LDS, [BM][CS_LD]) leaves as 16-byte row segments; CH = segments per row.
// The residual OR the per-batch row vector of a segment (the usual case: a layer has one of them) is requested for ALL
// of a thread's segments before the accumulators are staged (epi_prefetch), so that latency overlaps part 1; a layer
// with both reads the row vector in line.
// ROWS rows of the LDS tile belong to one epilogue pass; row r of the pass is tile row (r / SEG) * WM + row0 + r % SEG
// (SEG consecutive rows per wave-row; one pass: SEG = WM, row0 = 0 -> the identity).
// Branch-free on purpose (round 4): every condition folds into the buffer offset -- an out-of-range offset returns zeros --
// so the MAX_CH requests leave back to back.  With `if (...) pre[k] = *ptr` hipcc put every load into its own conditional
// block with s_waitcnt vmcnt(0) behind it: MAX_CH (and, for the bias, 4 x NI) SERIAL round trips per block, 11.8 us of the
// 72-us life of a 64x64-level conv3x3_halo_kernel block (tools/probes/gemm_timeline.py).  32-bit offsets: the host bounds the
// residual extent (plan_gemm).  enable = false: nothing is fetched (zeros).
typedef unsigned int vd_u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int vd_u32x2_t __attribute__((ext_vector_type(2)));
template <int ROWS, int CH, int NT, int MAX_CH, int SEG, int WM>
__device__ __forceinline__ void epi_prefetch(const EpiCtx& e, int M, int m0, int row0, int out_n0, int tid, uint4* pre, bool enable = true) {
    const bool vec_ok = ((e.N & 7) == 0) && ((e.ldr & 7) == 0);
    const bool want_res = (e.flags & VD_EPI_RESIDUAL) != 0;
    const bool want_rv = !want_res && (e.flags & VD_EPI_ROWVEC) != 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<f16*>(want_res ? e.res : e.rowvec), 0, (enable && vec_ok && (want_res || want_rv)) ? 0x7fffffff : 0, 0x00020000);
#pragma unroll
    for (int k = 0; k < MAX_CH; ++k) {
        const int c = tid + k * NT;
        const int r = c / CH, cc = (c % CH) * 8;
        const int row = m0 + (r / SEG) * WM + row0 + (r % SEG), col = out_n0 + cc;
        const bool ok = c < ROWS * CH && row < M && col + 8 <= e.N;
        const unsigned off = want_res ? (unsigned)((row * e.ldr + col) * 2) : (unsigned)(((row / e.rows_per_batch) * e.N + col) * 2);
        const vd_u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(ok ? off : OOB_OFFSET), 0, 0);
        pre[k] = make_uint4(v[0], v[1], v[2], v[3]);
    }
}

// bias of the 4 x NI column groups of a lane (columns col0 + j * 32 + 8 g .. + 4) and the LayerNorm-fold row statistics of its
// MI rows, all requested back to back (see epi_prefetch).  fast = N % 4 == 0 (else zeros: the caller reads the ragged bias
// element-wise).  Columns past N lie beyond the descriptor's range.
template <int NI>
__device__ __forceinline__ void epi_load_bias(const EpiCtx& e, int N, int col0, U2H4* b, bool enable) {
    const bool want = enable && (e.flags & VD_EPI_BIAS) && !(e.flags & VD_EPI_BIAS_ALONG_M) && (N & 3) == 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(e.bias), 0, want ? N * 2 : 0, 0x00020000);
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const vd_u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rs, (col0 + j * 32 + 8 * g) * 2, 0, 0);
            b[j * 4 + g].u = make_uint2(v[0], v[1]);
        }
}
// Row statistics of the LayerNorm fold, one 16-byte request per row whatever the source (no branch around the requests):
// (mean, rstd) fp32 from vd_row_stats_f16 in .x / .y, or -- VD_EPI_LN_SUMS -- the producer's row_sums, two int64 fixed-point
// sums (ln_sums_decode).
template <int MI>
__device__ __forceinline__ void epi_load_lnstats(const float* ln_stats, int M, int z, int row0, uint4* st, bool enable, bool sums, int rows_total) {
    const int rb = sums ? 16 : 8;
    // (mean, rstd) form: the 16-byte request of the LAST row reaches 8 bytes past the table; the descriptor covers them (+ 8) so the
    // result does not hang on how the range check treats a request that straddles the end -- the extra dwords are never used
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ln_stats), 0, (enable && ln_stats != nullptr) ? rows_total * rb + (sums ? 0 : 8) : 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int row = row0 + i * 32;
        const vd_u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(row < M ? (unsigned)((z * M + row) * rb) : OOB_OFFSET), 0, 0);
        st[i] = make_uint4(v[0], v[1], v[2], v[3]);   // (never __builtin_bit_cast a swizzle: clang reads element 0 for both)
    }
}
// VdGemmDesc.row_sums: fixed point so that the atomic accumulation is ORDER-INDEPENDENT (integer adds): the same inputs give
// the same bits in every run (fp32 atomics made two runs of one UNet forward differ by 1.4e-3 rel-L2 -- the fp16 rounding flips
// a 1e-7 perturbation of a LayerNorm triggers downstream).  sum x 2^24, sum of squares x 2^16 in int64: |row sum| < 2^38,
// sum of squares < 2^46 (1280 columns of |x| <= 65504 need 2^23 and 2^43).
constexpr float LN_SUM_SCALE = 16777216.0f, LN_SQ_SCALE = 65536.0f;
__device__ __forceinline__ void ln_sums_decode(const uint4 raw, int K, float eps, float& rstd, float& nmr) {
    const long long S1 = (long long)(((unsigned long long)raw.y << 32) | raw.x);
    const long long S2 = (long long)(((unsigned long long)raw.w << 32) | raw.z);
    // E[x^2] - mean^2 in double: the integers are exact, and in fp32 the one-pass form loses (mean / sigma)^2 * 6e-8 of the variance
    // for rows whose mean dwarfs their spread (a handful of operations per row and thread)
    const double inv_k = 1.0 / (double)K;
    const double mean = (double)S1 * (1.0 / (double)LN_SUM_SCALE) * inv_k;
    double var = (double)S2 * (1.0 / (double)LN_SQ_SCALE) * inv_k - mean * mean;
    if (var < 0.0) var = 0.0;
    rstd = rsqrtf((float)var + eps);
    nmr = -(float)mean * rstd;
}

// keep: the stored values are also written back into the tile (the statistics pass behind part 2 reads them there)
template <int ROWS, int CH, int NT, int MAX_CH, int CS_LD, int SEG, int WM>
__device__ __forceinline__ void epi_writeout(const EpiCtx& e, int M, int m0, int row0, int out_n0, int tid, f16* cs, const uint4* pre,
                                             bool nt, bool keep = false) {
    const bool vec_ok = ((e.N & 7) == 0) && ((e.ldr & 7) == 0) && ((e.ldc & 7) == 0);
    const bool both = (e.flags & VD_EPI_RESIDUAL) && (e.flags & VD_EPI_ROWVEC);
    if (vec_ok && !both) {
        // the usual case as a compact loop: the uniform decisions (alignment, store policy, write-back) are taken once, so the
        // unrolled iterations do not hop over the element-wise path they never take (instruction fetch: see conv_halo_kernel.h)
        auto fast = [&](auto nt_tag, auto keep_tag) {
            constexpr bool NTS = decltype(nt_tag)::value, KEEP = decltype(keep_tag)::value;
#pragma unroll
            for (int k = 0; k < MAX_CH; ++k) {
                const int c = tid + k * NT;
                const int r = c / CH, cc = (c % CH) * 8;
                const int row = m0 + (r / SEG) * WM + row0 + (r % SEG), col = out_n0 + cc;
                if (c < ROWS * CH && row < M && col < e.N) {
                    U4H8 t, a, o;
                    t.u = *reinterpret_cast<const uint4*>(cs + r * CS_LD + cc);
                    a.u = pre[k];
#pragma unroll
                    for (int i = 0; i < 8; ++i) o.e[i] = (f16)((float)t.e[i] + (float)a.e[i] + 0.f);
                    f16* dst = reinterpret_cast<f16*>(e.out) + (size_t)row * e.ldc + col;
                    if constexpr (NTS) vd_store16_nt(dst, o.u);
                    else *reinterpret_cast<uint4*>(dst) = o.u;
                    if constexpr (KEEP) *reinterpret_cast<uint4*>(cs + r * CS_LD + cc) = o.u;
                }
            }
        };
        // VdGemmDesc.row_sums: (sum, sum of squares) of the STORED values of each row over this block's columns, folded over the
        // CH (a power of two <= 64: the host asked vd_gemm_row_sums_ok) consecutive lanes that share the row, one pair of
        // atomic adds per row and block.  Its own loop (never taken by launches without the pointer: the usual path above keeps
        // its instruction footprint).
        if (e.row_sums != nullptr && (CH & (CH - 1)) == 0 && CH <= 64) {
#pragma unroll
            for (int k = 0; k < MAX_CH; ++k) {
                const int c = tid + k * NT;
                const int r = c / CH, cc = (c % CH) * 8;
                const int row = m0 + (r / SEG) * WM + row0 + (r % SEG), col = out_n0 + cc;
                const bool ok = c < ROWS * CH && row < M && col < e.N;
                float s1 = 0.f, s2 = 0.f;
                U4H8 o;
                if (ok) {
                    U4H8 t, a;
                    t.u = *reinterpret_cast<const uint4*>(cs + r * CS_LD + cc);
                    a.u = pre[k];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        o.e[i] = (f16)((float)t.e[i] + (float)a.e[i] + 0.f);
                        const float v = (float)o.e[i];
                        s1 += v;
                        s2 = fmaf(v, v, s2);
                    }
                    f16* dst = reinterpret_cast<f16*>(e.out) + (size_t)row * e.ldc + col;
                    if (nt) vd_store16_nt(dst, o.u);
                    else *reinterpret_cast<uint4*>(dst) = o.u;
                    if (keep) *reinterpret_cast<uint4*>(cs + r * CS_LD + cc) = o.u;
                }
#pragma unroll
                for (int sh = 1; sh < CH; sh <<= 1) {
                    s1 += __shfl_xor(s1, sh, 64);
                    s2 += __shfl_xor(s2, sh, 64);
                }
                if (ok && (c % CH) == 0) {   // fixed point: integer adds commute exactly (ln_sums_decode)
                    atomicAdd(e.row_sums + (size_t)row * 2, (unsigned long long)__float2ll_rn(s1 * LN_SUM_SCALE));
                    atomicAdd(e.row_sums + (size_t)row * 2 + 1, (unsigned long long)__float2ll_rn(s2 * LN_SQ_SCALE));
                }
            }
            return;
        }
        if (nt) {
            if (keep) fast(std::true_type{}, std::true_type{});
            else fast(std::true_type{}, std::false_type{});
        } else {
            if (keep) fast(std::false_type{}, std::true_type{});
            else fast(std::false_type{}, std::false_type{});
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < MAX_CH; ++k) {
        const int c = tid + k * NT;
        if (c < ROWS * CH) {
            const int r = c / CH, cc = (c % CH) * 8;
            const int row = m0 + (r / SEG) * WM + row0 + (r % SEG), col = out_n0 + cc;
            if (row < M && col < e.N) {
                U4H8 t;
                t.u = *reinterpret_cast<const uint4*>(cs + r * CS_LD + cc);
                if (vec_ok && col + 8 <= e.N) {
                    U4H8 a, b, o;
                    a.u = pre[k];
                    b.u = make_uint4(0, 0, 0, 0);
                    if (both) b.u = *reinterpret_cast<const uint4*>(e.rowvec + (size_t)(row / e.rows_per_batch) * e.N + col);
#pragma unroll
                    for (int i = 0; i < 8; ++i) o.e[i] = (f16)((float)t.e[i] + (float)a.e[i] + (float)b.e[i]);
                    // nt: streaming output that does not evict the weight / activation panels the other tiles of this
                    // XCD keep re-reading from its 4 MiB L2; the host decides per launch (GemmArgs.nt_store)
                    f16* dst = reinterpret_cast<f16*>(e.out) + (size_t)row * e.ldc + col;
                    if (nt) vd_store16_nt(dst, o.u);
                    else *reinterpret_cast<uint4*>(dst) = o.u;
                    if (keep) *reinterpret_cast<uint4*>(cs + r * CS_LD + cc) = o.u;
                } else {
                    float v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = (float)t.e[i];
                    epi_finish8(e, row, col, v);
                }
            }
        }
    }
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt range");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int BM, int BN, int NT, int STAGES, int KB>
constexpr int gemm_lds_bytes();   // stages / epilogue tile (defined with the launcher below)
// instances whose epilogue can emit per-channel statistics (VdGemmDesc.out_stats): one epilogue pass, the tile + the lane
// scratch fit the LDS the main loop owns anyway (occupancy unchanged), not a LayerNorm-fold variant
template <int BM, int BN, int NT, int STAGES, int KB, bool LNF>
constexpr bool gemm_emits_stats();

// MID (3 stages, 64-deep tiles only): the main loop of conv3x3_halo_kernel's MODE 2 -- the barrier of a K tile sits BEHIND
// its first two k-steps, tiles are requested two ahead, operand fragments one k-step ahead in two named register sets with
// the request / MFMA order pinned: nothing a wave needs right after the barrier depends on it.
template <int BM, int BN, int WM, int WN, int NT, int STAGES, int KB, int OCC, bool LNF = false, bool MID = false>
__global__ __launch_bounds__(NT, OCC) void gemm_f16_kernel(const GemmArgs p) {
    static_assert(!MID || (STAGES == 3 && KB == 64), "the mid-barrier loop walks 64-deep tiles through three stages");
    static_assert(KB == 64 || KB == 32, "K tile depth");
    static_assert(STAGES >= 2, "LDS ring needs at least two stages");
    constexpr int KROW_BYTES = KB * 2;  // one LDS row of a stage
    constexpr int SLOTS = KB / 8;       // 16-byte slots per row == threads cooperating on a row
    constexpr int KSUB = 64 / KB;       // kernel K tiles per planner K tile (the planner counts in 64s)
    constexpr int KS = KB / 16;         // MFMA k-steps per tile (even: fragment set of step ks is ks & 1)
    constexpr int WAVES_N = BN / WN;
    constexpr int WAVES_M = BM / WM;
    static_assert(WAVES_M * WAVES_N * 64 == NT, "waves must tile the block");
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int NMF = MI * NI;     // MFMAs per k-step and wave
    constexpr int RPP = NT / SLOTS;  // rows staged per pass: SLOTS threads per row
    // a staging pass covers RPP rows; tiles that are not a multiple of it (128x320 / 128x160 with 32-deep K tiles) carry
    // padding rows in LDS, which the last pass fills with zeros (out-of-range source offset)
    constexpr int A_PASSES = (BM + RPP - 1) / RPP, B_PASSES = (BN + RPP - 1) / RPP;
    constexpr int BMP = A_PASSES * RPP, BNP = B_PASSES * RPP;
    constexpr int LPT = A_PASSES + B_PASSES;  // DMA pieces per thread and tile
    constexpr int STAGE_BYTES = (BMP + BNP) * KROW_BYTES;
    constexpr int CS_LD = BN + 8;  // fp16 epilogue tile leading dimension (halfs); row stride = odd multiple of 16 B
    // the epilogue tile goes through LDS in EP passes of BM / EP rows (2 where the whole tile would not fit: 256 x 320)
    constexpr int EP = (BM * CS_LD * 2 + BM * 8 > 160 * 1024) ? 2 : 1;
    static_assert((WM / 32) % EP == 0, "epilogue passes must divide the wave tile's 32-row blocks");
    constexpr int EPI_BYTES = (BM / EP) * CS_LD * 2;
    constexpr int D = STAGES - 1;  // prefetch distance in tiles
    static_assert(LPT * (D > 1 ? D - 1 : 1) < 64, "vmcnt range");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    VD_TL_DECL;
    VD_TL(0);   // block start

    const VdGemmDesc& d = p.d;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int hi = lane >> 5, l31 = lane & 31;

    // ---- XCD-aware tile mapping: block b runs on XCD b%8; give each XCD a contiguous run of logical tiles.  n fastest:
    // neighbouring tiles share the A row-panel in that XCD's L2 and every XCD streams ALL of W through the fabric (8 x W).
    // m fastest (p.mfast, chosen by the host where 8 x W outweighs tiles_n x A: the weight-heavy 16x16 / 32x32 levels): an
    // XCD owns a few weight column panels and re-reads the (small) activation instead.
    const int ntiles = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = p.mfast ? bid % p.tiles_m : bid / p.tiles_n;
    const int tn = p.mfast ? bid / p.tiles_m : bid - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int split = blockIdx.y;
    const int z = blockIdx.z;

    // Short-lived blocks (the small tiles of the K = 320 .. 1280 projections): the epilogue's operands -- bias, LayerNorm row
    // statistics, residual / row-vector segments -- are requested HERE, in front of the first K tile, not behind the main loop.
    // Their round trips were 2.0-2.8 us of a 9-16 us block life (tools/probes/gemm_timeline.py: main loop done -> tile in
    // LDS); now they overlap the prologue's.  They are the oldest requests of the wave and loads return in order, so the
    // counted vmcnt waits of the loop hold unchanged.  Blocks that leave fp32 partials never look at them.
    constexpr int MAX_CH_H = BM * (BN / 8) / NT;
    // Only where the ~18 registers they occupy across the main loop cost no occupancy: one MFMA tile per wave, no LayerNorm
    // fold (128x64 on 8 waves with the fold: 78 -> 91 registers, 6 -> 5 waves per SIMD; 128x128 on 8 waves spilled).
    constexpr bool HOIST = !MID && !LNF && EP == 1 && MI * NI == 1 && MAX_CH_H <= 2;
    const EpiCtx e = make_epi(d, z);
    const bool hoisted = HOIST && d.act != VD_ACT_GEGLU && p.hoist;
    uint4 pre_h[HOIST ? MAX_CH_H : 1];
    constexpr bool BIAS_ALL = NI <= 2;       // wider wave tiles request the bias per 32-column group (4 requests at a time)
    U2H4 bias_r[BIAS_ALL ? NI * 4 : 4];
    uint4 lnst_r[MI];
    const bool ln_sums = (d.flags & VD_EPI_LN_SUMS) != 0;
    if constexpr (HOIST) {
        epi_prefetch<BM, BN / 8, NT, MAX_CH_H, WM, WM>(e, d.M, m0, 0, n0, tid, pre_h, hoisted);
        epi_load_bias<NI>(e, d.N, n0 + wn * WN + 4 * hi, bias_r, hoisted);
        if constexpr (LNF) epi_load_lnstats<MI>(d.ln_stats, d.M, z, m0 + wm * WM + l31, lnst_r, hoisted, ln_sums, d.M * d.batch);
    }

    // LayerNorm fold: the block's BN entries of colsum wait in LDS behind the stages / the epilogue tile, so the epilogue's
    // register phase reads them with ds_read instead of one more global round trip per column group
    if constexpr (LNF) {
        float* ln_cs = reinterpret_cast<float*>(smem + gemm_lds_bytes<BM, BN, NT, STAGES, KB>());
        for (int c = tid * 4; c < BN; c += NT * 4) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n0 + c + 4 <= d.N) v = *reinterpret_cast<const float4*>(d.colsum + n0 + c);
            else {
                float t[4] = {0.f, 0.f, 0.f, 0.f};
                for (int q = 0; q < 4; ++q)
                    if (n0 + c + q < d.N) t[q] = d.colsum[n0 + c + q];
                v = make_float4(t[0], t[1], t[2], t[3]);
            }
            *reinterpret_cast<float4*>(ln_cs + c) = v;
        }
    }

    // ---- operands are read through buffer descriptors: one SGPR base + per-lane 32-bit byte offset + a scalar
    // K offset per tile.  Out-of-image taps, rows >= M / N and the ragged K tail simply use an out-of-range
    // offset and the hardware returns zeros: no branches and (within a conv tap) no VALU work per K tile.
    const i32x4 ws_a0 = make_rsrc_words(reinterpret_cast<const f16*>(d.a0) + (size_t)z * d.stride_a, p.a0_bytes);
    const i32x4 ws_a1 = make_rsrc_words(d.a1 ? reinterpret_cast<const f16*>(d.a1) + (size_t)z * d.stride_a : d.a0,
                                        d.a1 ? p.a1_bytes : 0u);
    const i32x4 ws_w = make_rsrc_words(reinterpret_cast<const f16*>(d.w) + (size_t)z * d.stride_w, p.w_bytes);

    // per-thread gather coordinates: SLOTS threads per LDS row, RPP rows per pass
    const int lrow = tid / SLOTS, lslot = tid % SLOTS;
    int a_iy0[A_PASSES], a_ix0[A_PASSES], a_pix[A_PASSES];
    const int HWo = d.Hout * d.Wout;
    const int Hv = d.Hin << d.ups, Wv = d.Win << d.ups;
#pragma unroll
    for (int ps = 0; ps < A_PASSES; ++ps) {
        const int m = (BMP == BM || lrow + RPP * ps < BM) ? m0 + lrow + RPP * ps : d.M;  // padding rows: out of range
        if (m < d.M && p.plain) {  // plain matrix / 1x1 stride-1 conv: output row == input pixel, no index division
            a_iy0[ps] = 0;
            a_ix0[ps] = 0;
            a_pix[ps] = m;
        } else if (m < d.M) {
            const int b = m / HWo;
            const int rem = m - b * HWo;
            const int oy = rem / d.Wout;
            const int ox = rem - oy * d.Wout;
            a_iy0[ps] = oy * d.stride - d.pad;
            a_ix0[ps] = ox * d.stride - d.pad;
            a_pix[ps] = b * d.Hin * d.Win;
        } else {
            a_iy0[ps] = -(1 << 28);  // always out of bounds -> zero rows
            a_ix0[ps] = 0;
            a_pix[ps] = 0;
        }
    }

    const int ctot = d.c0 + d.c1;
    int kt_end = (split + 1) * p.kt_per_split;
    if (kt_end > p.kt_total) kt_end = p.kt_total;
    const int kt0 = split * p.kt_per_split * KSUB;  // in units of this kernel's K tile
    const int nk = kt_end * KSUB - kt0;
    const bool ragged = (d.K % BK) != 0;

    // ---- LDS-DMA issue state.  The DMA destination is lane-linear (wave-uniform base + lane * 16), so the XOR swizzle
    // of the LDS image is applied on the SOURCE side: the lane fetches the logical slot that lives at its physical slot.
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned wave_dst = (unsigned)(__builtin_amdgcn_readfirstlane(wave) * 1024);  // one wave-wide DMA = 64 x 16 bytes
    const int sw_slot = lslot ^ lds_swz<KB>(lrow);
    unsigned dvoff_b[B_PASSES];
#pragma unroll
    for (int ps = 0; ps < B_PASSES; ++ps) {
        const int n = (BNP == BN || lrow + RPP * ps < BN) ? n0 + lrow + RPP * ps : d.N;
        dvoff_b[ps] = (n < d.N) ? (unsigned)((n * d.ldw + sw_slot * 8) * 2) : OOB_OFFSET;
    }
    unsigned dvoff_a[A_PASSES];
    // (tap, channel offset) of the next tile to issue, advanced incrementally: no per-tile integer division
    int n_tap = (kt0 * KB) / ctot;
    int n_cc = kt0 * KB - n_tap * ctot;
    int n_ky = n_tap / d.ksize, n_kx = n_tap - n_ky * d.ksize;
    bool seg_dirty = true;
    bool n_second = false;
    // state of the tile being issued (wave-uniform)
    i32x4 is_src = ws_a0;
    unsigned is_soff_a = 0, is_soff_b = 0, is_dst_a = 0, is_dst_b = 0;
    unsigned is_kmask = 0;  // 0x80000000 on lanes whose slot lies beyond a ragged K: pushes the offset out of range -> zeros
    auto issue_begin = [&](int t, int buf) {
        const bool second = n_cc >= d.c0;
        if (second != n_second) { n_second = second; seg_dirty = true; }
        if (seg_dirty) {  // wave-uniform: first tile, new tap, or switch to the concatenated source
            seg_dirty = false;
            const int ld = second ? d.lda1 : d.lda0;
#pragma unroll
            for (int ps = 0; ps < A_PASSES; ++ps) {
                const int iy = a_iy0[ps] + n_ky, ix = a_ix0[ps] + n_kx;
                const bool ok = ((unsigned)iy < (unsigned)Hv) && ((unsigned)ix < (unsigned)Wv);
                const int pix = a_pix[ps] + (iy >> d.ups) * d.Win + (ix >> d.ups);
                dvoff_a[ps] = ok ? (unsigned)((pix * ld + sw_slot * 8) * 2) : OOB_OFFSET;
            }
        }
        const int kglob = t * KB;
        is_src = second ? ws_a1 : ws_a0;
        is_soff_a = (unsigned)((second ? n_cc - d.c0 : n_cc) * 2);
        is_soff_b = (unsigned)(kglob * 2);
        is_dst_a = lds0 + (unsigned)(buf * STAGE_BYTES) + wave_dst;
        is_dst_b = is_dst_a + BMP * KROW_BYTES;
        is_kmask = (ragged && (kglob + sw_slot * 8 >= d.K)) ? OOB_OFFSET : 0u;
        n_cc += KB;  // advance to the next K tile
        if (n_cc >= ctot) {
            n_cc -= ctot;
            ++n_tap;
            ++n_kx;
            if (n_kx == d.ksize) { n_kx = 0; ++n_ky; }
            seg_dirty = true;
        }
    };
    auto issue_piece = [&](int pc) {  // pc is a compile-time constant at every call site (unrolled loops)
        if (pc < A_PASSES) {
            dma16(is_src, is_dst_a + pc * RPP * KROW_BYTES, dvoff_a[pc < A_PASSES ? pc : 0] | is_kmask, is_soff_a);
        } else {
            const int q = pc - A_PASSES;
            dma16(ws_w, is_dst_b + q * RPP * KROW_BYTES, dvoff_b[q < B_PASSES ? q : 0] | is_kmask, is_soff_b);
        }
    };

    // acc[i][j] holds the TRANSPOSED 32x32 sub-tile (MFMA A operand = W rows, B operand = activation rows):
    // lane owns output row m = l31 and, per register group g = r>>2, four consecutive columns n = 8g + 4hi + (r&3).
    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // LDS read offsets: the swizzle key is the same for every 32-row fragment of a lane, so each lane needs one offset
    // per k-step and operand; fragments are immediate offsets (32 rows).
    int rd_a[KS], rd_b[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        rd_a[ks] = lds_off_kb<KB>(wm * WM + l31, ks * 2 + hi);
        rd_b[ks] = BMP * KROW_BYTES + lds_off_kb<KB>(wn * WN + l31, ks * 2 + hi);
    }
    // LayerNorm fold without a statistics pass (d.ln_stats == NULL): every A fragment passes through read_frags exactly once,
    // and a lane's fragments all belong to ONE row per 32-row block (row l31, k-half hi) -- sum and sum of squares of that row
    // are two v_dot2_f32_f16 per register on the way (8 VALU instructions per fragment, issued beside the MFMAs), the other
    // half of the row is one lane^32 exchange in the epilogue.  (Round 2 did this with converts + FMAs, 3x the instructions,
    // and it cost as much as the statistics launches it removed.)
    const bool ln_inloop = LNF && d.ln_stats == nullptr;
    float ls1[MI], ls2[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) ls1[i] = ls2[i] = 0.f;
    auto read_frags = [&](const char* st, int ks, f16x8* a, f16x8* b) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            U4H8 t;
            t.u = *reinterpret_cast<const uint4*>(st + rd_a[ks] + i * 32 * KROW_BYTES);
            a[i] = t.h;
            if constexpr (LNF) {
                if (ln_inloop) {
                    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
                    const f16x2_t one2 = {(_Float16)1.0f, (_Float16)1.0f};
                    const f16x2_t* pr = reinterpret_cast<const f16x2_t*>(&t);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        ls1[i] = __builtin_amdgcn_fdot2(pr[q], one2, ls1[i], false);
                        ls2[i] = __builtin_amdgcn_fdot2(pr[q], pr[q], ls2[i], false);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            U4H8 t;
            t.u = *reinterpret_cast<const uint4*>(st + rd_b[ks] + j * 32 * KROW_BYTES);
            b[j] = t.h;
        }
    };

    // LayerNorm fold (VD_EPI_LNFOLD): the epilogue applies rstd * acc - rstd * mean * colsum with the row statistics of A
    // read from d.ln_stats (vd_row_stats_f16).  Round 2 first accumulated them inside the K loop from the LDS-resident A
    // tiles (no extra pass over A at all), but every column block of a row panel repeats that work and the loop of the
    // short-K projections has no slack for it: +24 us on the 57 us q/k/v projection of the 64x64 level, +35 us on its
    // GEGLU projection -- as much as the LayerNorm launches the fold removes.  LNF stays a compile-time property of the
    // instantiation so that launches without it carry none of this.
    if constexpr (MID) {
        if (nk > 0) {
            auto issue_tile = [&](int t, int buf) {
                issue_begin(kt0 + t, buf);
#pragma unroll
                for (int pc = 0; pc < LPT; ++pc) issue_piece(pc);
            };
            auto mma = [&](const f16x8* a, const f16x8* b) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int nj = 0; nj < NI; ++nj)
                        acc[mi][nj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[nj], a[mi], acc[mi][nj], 0, 0, 0);
            };
            auto pin = []() { __builtin_amdgcn_sched_barrier(0); };
            issue_tile(0, 0);
            if (nk > 1) issue_tile(1, 1);
            wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            f16x8 a0f[MI], b0f[NI], a1f[MI], b1f[NI];
            read_frags(smem, 0, a0f, b0f);
            int cb = 0;
            for (int i = 0; i < nk; ++i) {
                const int nb = cb == 2 ? 0 : cb + 1, ib = nb == 2 ? 0 : nb + 1;
                const char* st = smem + cb * STAGE_BYTES;
                read_frags(st, 1, a1f, b1f);
                pin();
                mma(a0f, b0f);
                pin();
                read_frags(st, 2, a0f, b0f);
                pin();
                mma(a1f, b1f);
                // tile i + 1 (requested one tile ago) has landed for every wave; every wave has left tile i - 1, whose
                // stage the request below overwrites
                wait_vm<0>();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (i + 2 < nk) issue_tile(i + 2, ib);
                read_frags(st, 3, a1f, b1f);
                pin();
                mma(a0f, b0f);
                pin();
                if (i + 1 < nk) read_frags(smem + nb * STAGE_BYTES, 0, a0f, b0f);
                pin();
                mma(a1f, b1f);
                cb = nb;
            }
        }
    } else {
    // ---- main loop: every K tile = { wait for the tile, barrier, issue the tile D ahead as ONE burst, then the
    // compiler-scheduled ds_read / MFMA stream of the landed tile }.  The burst gives a DMA the longest possible lead (a
    // whole iteration per stage of distance), and hipcc's own interleave of the fragment reads with the MFMAs (fine
    // lgkmcnt ladders) beats a hand-pinned order: a software-pipelined variant (fragments double-buffered across the barrier,
    // DMA pieces in the MFMA gaps, order pinned with sched_barrier; -DVD_GEMM_PIPELINED of rounds 1-5, removed in round 6)
    // measured 4 % SLOWER over the UNet forward on the same box (13.50 vs 12.95 ms; again 10.43 vs 10.50 in round 5) although
    // it won by 3-4 % on a back-to-back micro-benchmark of one shape.
    if (nk > 0) {
#pragma unroll
        for (int j = 0; j < D; ++j)
            if (j < nk) {
                issue_begin(kt0 + j, j);
#pragma unroll
                for (int pc = 0; pc < LPT; ++pc) issue_piece(pc);
            }
        int cbuf = 0, ibuf = D % STAGES;
        for (int i = 0; i < nk; ++i) {
            // tile i must have landed; in steady state the D-1 younger tiles stay in flight across the barrier
            if (i + D - 1 < nk) wait_vm<LPT * (D - 1)>();
            else wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");  // LDS reads below must not be hoisted above the barrier
#ifdef VD_TIMELINE
            if (i == 0) VD_TL(1);   // first K tile landed for every wave
#endif
            if (i + D < nk) {
                issue_begin(kt0 + i + D, ibuf);
#pragma unroll
                for (int pc = 0; pc < LPT; ++pc) issue_piece(pc);
            }
            const char* st = smem + cbuf * STAGE_BYTES;
    #pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                f16x8 af[MI], bf[NI];
                read_frags(st, ks, af, bf);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int nj = 0; nj < NI; ++nj)
                        acc[mi][nj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[nj], af[mi], acc[mi][nj], 0, 0, 0);
            }
            cbuf = (cbuf + 1 == STAGES) ? 0 : cbuf + 1;
            ibuf = (ibuf + 1 == STAGES) ? 0 : ibuf + 1;
        }
    }
    }   // !MID
    constexpr bool lnf = LNF;
    wait_vm<0>();
    __syncthreads();  // every wave is done with the stages: the epilogue tile re-uses that LDS
    VD_TL(2);   // main loop done

    // ---- split-K with arrival counters: every block of a tile leaves its fp32 accumulators in the workspace in REGISTER
    // order ([8-byte group][thread]: fully coalesced, no address arithmetic per element); the block that arrives last sums
    // the slabs in split order (its own included, so the result does not depend on which block that is) and carries on
    // into the fused epilogue.  Saves the reduce launch and its drain / fill between two kernels.
    // Coherence: the blocks of a tile run on different XCDs, whose L2s are not coherent for plain accesses.  An agent-scope
    // fence pair would write back and invalidate the whole L2 per block (measured: +80 us per launch); instead every slab
    // access is itself an agent-scope relaxed atomic (sc1: write-through / read-through at the device coherence point),
    // ordered against the arrival counter by s_waitcnt vmcnt(0) + the workgroup barrier.
    if (gridDim.y > 1 && d.sync != nullptr) {
        typedef unsigned long long u64;
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        constexpr int G2 = MI * NI * 8;                     // 8-byte groups per thread
        const int nsplit = gridDim.y;
        const size_t tile_id = ((size_t)z * p.tiles_m + tm) * p.tiles_n + tn;
        u64* slab = reinterpret_cast<u64*>(d.ws) + tile_id * nsplit * (size_t)(G2 * NT);
        u64* mine = slab + (size_t)split * (G2 * NT) + tid;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int g = 0; g < 8; ++g)
                    __hip_atomic_store(mine + ((i * NI + j) * 8 + g) * NT,
                                       __builtin_bit_cast(u64, f32x2{acc[i][j][g * 2], acc[i][j][g * 2 + 1]}),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's slab stores are acknowledged at device scope
        __syncthreads();                                    // ... and every wave's
        int* flag = reinterpret_cast<int*>(smem);
        if (tid == 0) {
            const int last = (__hip_atomic_fetch_add(&d.sync[tile_id], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nsplit - 1) ? 1 : 0;
            if (last)   // every block of the tile has arrived: re-arm for the next launch
                __hip_atomic_store(&d.sync[tile_id], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *flag = last;
        }
        __syncthreads();
        const int last = *flag;
        if (!last) return;
        __syncthreads();                                    // flag has been read: the LDS is free for the epilogue tile
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;
        const u64* src = slab + tid;
        for (int s = 0; s < nsplit; ++s, src += G2 * NT) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    u64 v[8];
#pragma unroll
                    for (int g = 0; g < 8; ++g)
                        v[g] = __hip_atomic_load(src + ((i * NI + j) * 8 + g) * NT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int g = 0; g < 8; ++g) {
                        const f32x2 f = __builtin_bit_cast(f32x2, v[g]);
                        acc[i][j][g * 2] += f.x;
                        acc[i][j][g * 2 + 1] += f.y;
                    }
                }
        }
    }

    // ---- split-K (slabs for the reduce kernel) / fp32 output: straight from registers (4 consecutive floats per lane and group)
    if ((gridDim.y > 1 && d.sync == nullptr) || (d.flags & VD_EPI_OUT_F32)) {
        const bool partial = gridDim.y > 1;
        float* base = partial ? d.ws + ((size_t)z * gridDim.y + split) * (size_t)d.M * d.N
                              : reinterpret_cast<float*>(e.out);
        const int ld = partial ? d.N : e.ldc;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int row = m0 + wm * WM + i * 32 + l31;
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = n0 + wn * WN + j * 32 + 8 * g + 4 * hi;
                    if (row < d.M && col < d.N) {
                        float v[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = acc[i][j][g * 4 + q];
                        if (!partial) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                float t = v[q];
                                if ((e.flags & VD_EPI_BIAS) && col + q < d.N)
                                    t += (float)((e.flags & VD_EPI_BIAS_ALONG_M) ? e.bias[row] : e.bias[col + q]);
                                v[q] = apply_act(e.act, t) * e.alpha;
                            }
                        }
                        float* o = base + (size_t)row * ld + col;
                        if (col + 4 <= d.N && (ld & 3) == 0) {
                            *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
                        } else {
                            // ragged edge, element by element.  volatile keeps this path apart from the 16-byte store above:
                            // hipcc otherwise merges the two (one dwordx3 + a shared conditional dword store per group,
                            // i.e. twice the store instructions and no full 16-byte writes; +4 us per split-K launch)
                            volatile float* ov = o;
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                if (col + q < d.N) ov[q] = v[q];
                        }
                    }
                }
        }
        VD_TL(4);
        VD_TL_FLUSH(p.tl);
        return;
    }

    // ---- fused epilogue, part 1 (registers): (LayerNorm fold) + bias -> act / GEGLU -> * alpha -> fp16 into an LDS
    // tile [BM][OUT_N]
    f16* cs = reinterpret_cast<f16*>(smem);
    const bool geglu = (d.act == VD_ACT_GEGLU);
    const bool want_stats = gemm_emits_stats<BM, BN, NT, STAGES, KB, LNF>() && d.out_stats != nullptr && p.stat_rows > 0;
    const int out_n0 = geglu ? tn * (BN / 2) : n0;

    const float* ln_cs = reinterpret_cast<const float*>(smem + gemm_lds_bytes<BM, BN, NT, STAGES, KB>());   // block-local colsum

    // residual / row-vector segments of part 2 are requested NOW so their latency overlaps part 1 (the block is
    // short-lived on the K = 320..1280 projections: every serial memory round trip shows)
    constexpr int MIP = MI / EP;          // 32-row blocks of a wave tile per pass
    constexpr int SEG = MIP * 32;         // consecutive tile rows per wave-row in one pass
    constexpr int PROWS = BM / EP;        // rows of the LDS tile
    constexpr int MAX_CH = PROWS * (BN / 8) / NT;
#pragma unroll
    for (int ep = 0; ep < EP; ++ep) {
    if (ep > 0) __syncthreads();  // the previous pass has left the LDS tile
    uint4 pre[MAX_CH];
    bool have_pre = false;
    if constexpr (HOIST) {
        if (hoisted) {
#pragma unroll
            for (int k = 0; k < MAX_CH; ++k) pre[k] = pre_h[k];
            have_pre = true;
        }
    }
    if (!have_pre) {
        if (geglu) epi_prefetch<PROWS, BN / 16, NT, MAX_CH, SEG, WM>(e, d.M, m0, ep * SEG, out_n0, tid, pre);
        else epi_prefetch<PROWS, BN / 8, NT, MAX_CH, SEG, WM>(e, d.M, m0, ep * SEG, out_n0, tid, pre);
        if (ep == 0) {   // bias / LayerNorm statistics of the whole wave tile, requested back to back behind the segments
            if constexpr (BIAS_ALL) epi_load_bias<NI>(e, d.N, n0 + wn * WN + 4 * hi, bias_r, true);
            if constexpr (LNF) epi_load_lnstats<MI>(d.ln_stats, d.M, z, m0 + wm * WM + l31, lnst_r, true, ln_sums, d.M * d.batch);
        }
    }
    const bool bias_fast = (d.N & 3) == 0;
#pragma unroll
    for (int ii = 0; ii < MIP; ++ii) {
        const int i = ep * MIP + ii;
        const int lrow_t = wm * WM + i * 32 + l31;        // row inside the block tile
        const int lrow_c = wm * SEG + ii * 32 + l31;      // row inside this pass's LDS tile
        const int row = m0 + lrow_t;
        float bm = 0.f;
        if ((e.flags & VD_EPI_BIAS) && (e.flags & VD_EPI_BIAS_ALONG_M) && row < d.M) bm = (float)e.bias[row];
        float ln_rstd = 1.f, ln_nmr = 0.f;  // y = rstd * acc - (mean * rstd) * colsum[n] + bias'[n]
        if (lnf) {
            if (ln_inloop) {
                const float s1 = ls1[i] + __shfl_xor(ls1[i], 32, 64), s2 = ls2[i] + __shfl_xor(ls2[i], 32, 64);
                const float inv_k = 1.0f / (float)d.K;
                const float mean = s1 * inv_k;
                float var = s2 * inv_k - mean * mean;
                if (var < 0.f) var = 0.f;
                ln_rstd = rsqrtf(var + d.ln_eps);
                ln_nmr = -mean * ln_rstd;
            } else {
                const uint4 st = lnst_r[i];
                if (ln_sums) {   // (sum, sum of squares) from the producer's row_sums
                    ln_sums_decode(st, d.K, d.ln_eps, ln_rstd, ln_nmr);
                } else {
                    ln_rstd = __uint_as_float(st.y);
                    ln_nmr = -__uint_as_float(st.x) * ln_rstd;
                }
            }
        }
        if (geglu) {
            // weight rows are packed per 64-row group as [32 value rows | 32 gate rows]: of each pair of 32-column MFMA
            // tiles of the wave, j = 2jj holds the values and j = 2jj + 1 the gates of output columns jj*32 ..
            if constexpr (NI % 2 == 0) {
#pragma unroll
                for (int jj = 0; jj < NI / 2; ++jj) {
                U2H4 bias_v[4], bias_g[4];
                if constexpr (!BIAS_ALL) {
                    epi_load_bias<1>(e, d.N, n0 + wn * WN + jj * 64 + 4 * hi, bias_v, true);
                    epi_load_bias<1>(e, d.N, n0 + wn * WN + jj * 64 + 32 + 4 * hi, bias_g, true);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int lc = wn * (WN / 2) + jj * 32 + 8 * g + 4 * hi;   // column inside the block's output tile
                    const int pn = n0 + wn * WN + jj * 64 + 8 * g + 4 * hi;    // packed weight row of the value element
                    U2H4 bv, bg, o;
                    if constexpr (BIAS_ALL) {
                        bv = bias_r[(2 * jj) * 4 + g];   // packed rows pn .. and pn + 32 ..: column groups 2 jj and 2 jj + 1 of the lane
                        bg = bias_r[(2 * jj + 1) * 4 + g];
                    } else {
                        bv = bias_v[g];
                        bg = bias_g[g];
                    }
                    if ((e.flags & VD_EPI_BIAS) && !bias_fast) {   // packed GEGLU weights have N % 128 == 0: never taken
                        bv.u = *reinterpret_cast<const uint2*>(e.bias + pn);
                        bg.u = *reinterpret_cast<const uint2*>(e.bias + pn + 32);
                    }
                    float4 cv = make_float4(0.f, 0.f, 0.f, 0.f), cg = cv;
                    if (lnf) {
                        cv = *reinterpret_cast<const float4*>(ln_cs + (pn - n0));
                        cg = *reinterpret_cast<const float4*>(ln_cs + (pn - n0) + 32);
                    }
                    const float cva[4] = {cv.x, cv.y, cv.z, cv.w}, cga[4] = {cg.x, cg.y, cg.z, cg.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float v = fmaf(ln_rstd, acc[i][2 * jj][g * 4 + q], ln_nmr * cva[q]) + (float)bv.e[q];
                        const float gt = fmaf(ln_rstd, acc[i][2 * jj + 1][g * 4 + q], ln_nmr * cga[q]) + (float)bg.e[q];
                        o.e[q] = (f16)(v * vd_gelu_erf(gt) * e.alpha);
                    }
                    *reinterpret_cast<uint2*>(cs + lrow_c * CS_LD + lc) = o.u;
                }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                U2H4 bias_j[4];
                if constexpr (!BIAS_ALL) epi_load_bias<1>(e, d.N, n0 + wn * WN + j * 32 + 4 * hi, bias_j, true);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int lc = wn * WN + j * 32 + 8 * g + 4 * hi;
                    const int col = n0 + lc;
                    float bq[4] = {bm, bm, bm, bm};
                    if ((e.flags & VD_EPI_BIAS) && !(e.flags & VD_EPI_BIAS_ALONG_M)) {
                        U2H4 t;
                        if constexpr (BIAS_ALL) t = bias_r[j * 4 + g];
                        else t = bias_j[g];
                        if (!bias_fast) t = epi_load_bias4(e, d.N, col);   // N % 4 != 0: element-wise
#pragma unroll
                        for (int q = 0; q < 4; ++q) bq[q] = (float)t.e[q];
                    }
                    if (lnf) {   // columns past N hold zeros
                        const float4 c4 = *reinterpret_cast<const float4*>(ln_cs + lc);
                        bq[0] = fmaf(ln_nmr, c4.x, bq[0]);
                        bq[1] = fmaf(ln_nmr, c4.y, bq[1]);
                        bq[2] = fmaf(ln_nmr, c4.z, bq[2]);
                        bq[3] = fmaf(ln_nmr, c4.w, bq[3]);
                    }
                    U2H4 o;
                    if (e.act == VD_ACT_NONE) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) o.e[q] = (f16)(fmaf(ln_rstd, acc[i][j][g * 4 + q], bq[q]) * e.alpha);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            o.e[q] = (f16)(apply_act(e.act, fmaf(ln_rstd, acc[i][j][g * 4 + q], bq[q])) * e.alpha);
                    }
                    *reinterpret_cast<uint2*>(cs + lrow_c * CS_LD + lc) = o.u;
                }
            }
        }
    }
    __syncthreads();
    VD_TL(3);   // epilogue tile in LDS

    // ---- part 2: coalesced 16-byte row segments: (+ rowvec) (+ residual) -> global
    if (geglu) epi_writeout<PROWS, BN / 16, NT, MAX_CH, CS_LD, SEG, WM>(e, d.M, m0, ep * SEG, out_n0, tid, cs, pre, p.nt_store != 0);
    else epi_writeout<PROWS, BN / 8, NT, MAX_CH, CS_LD, SEG, WM>(e, d.M, m0, ep * SEG, out_n0, tid, cs, pre, p.nt_store != 0, want_stats);
    }  // epilogue pass

    // ---- per-channel statistics of the stored tile for a consuming GroupNorm (plan_gemm admits the request only for plain
    // fp16 epilogues, vector-aligned outputs and tiles whose rows split into whole blocks of p.stat_rows rows of one image)
    if constexpr (gemm_emits_stats<BM, BN, NT, STAGES, KB, LNF>()) {
        if (want_stats) {
            __syncthreads();
            const int R = p.stat_rows, nsub = BM / R;
            const int left = (d.M - m0) / R;   // partials of this tile that lie inside the matrix
            emit_chan_stats<BN, CS_LD, NT>(cs, reinterpret_cast<float*>(smem + BM * CS_LD * 2), tid, R, nsub, left < nsub ? left : nsub,
                                           d.out_stats, (size_t)(m0 / R), d.N, n0, reinterpret_cast<unsigned long long*>(d.stat_sums), d.stat_img_rows);
        }
    }
    VD_TL(4);   // output stores issued (the flush waits for them: stamp 6 = stores acknowledged)
    VD_TL_FLUSH(p.tl);
}

template <int BM, int BN, int NT, int STAGES, int KB>
constexpr int gemm_lds_bytes() {
    constexpr int rpp = NT / (KB / 8);
    constexpr int stage = ((BM + rpp - 1) / rpp + (BN + rpp - 1) / rpp) * rpp * KB * 2 * STAGES;
    constexpr int ep = (BM * (BN + 8) * 2 + BM * 8 > 160 * 1024) ? 2 : 1;
    constexpr int epi = (BM / ep) * (BN + 8) * 2 + BM * 8;  // epilogue tile (one pass) + LayerNorm-fold row statistics
    return stage > epi ? stage : epi;
}

template <int BM, int BN, int NT, int STAGES, int KB, bool LNF>
constexpr bool gemm_emits_stats() {
    return !LNF && NT >= 256 && (BM * (BN + 8) * 2 + BM * 8 <= 160 * 1024) && stat_lds_bytes(BM, BN) <= gemm_lds_bytes<BM, BN, NT, STAGES, KB>();
}

template <int BM, int BN, int WM, int WN, int NT, int STAGES, int KB, int OCC, bool LNF = false, bool MID = false>
int launch_cfg(const GemmArgs& a, int nsplit, hipStream_t stream) {
    if (a.d.out_stats != nullptr && nsplit == 1 && !gemm_emits_stats<BM, BN, NT, STAGES, KB, LNF>()) {
        vd_set_error("vd_gemm_f16: this tile configuration cannot emit out_stats (ask vd_gemm_stat_rows first)");
        return VD_ERR_UNSUPPORTED;
    }
    constexpr int LDS = gemm_lds_bytes<BM, BN, NT, STAGES, KB>() + (LNF ? BN * 4 : 0);   // + the block's colsum entries
    static_assert(LDS <= 160 * 1024, "tile does not fit the CU's LDS");
    // the dynamic-LDS attribute is per device: one bit per device ordinal, set idempotently (safe under concurrent callers)
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16_kernel<BM, BN, WM, WN, NT, STAGES, KB, OCC, LNF, MID>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) {
            vd_set_error("vd_gemm_f16: cannot reserve %d bytes of LDS: %s", LDS, hipGetErrorString(e));
            return VD_ERR_LAUNCH;
        }
        done.fetch_or(bit, std::memory_order_release);
    }
    dim3 grid(a.tiles_m * a.tiles_n, nsplit, a.d.batch > 0 ? a.d.batch : 1);
    hipLaunchKernelGGL((gemm_f16_kernel<BM, BN, WM, WN, NT, STAGES, KB, OCC, LNF, MID>), grid, dim3(NT), LDS, stream, a);
    return vd_check_launch("vd_gemm_f16");
}

}  // namespace

// one-wave-per-SIMD instances (gemm_big.hip); cfg = TileCfg value
