// Halo-resident 3x3 convolution (stride 1, pad 1, optional nearest-2x upsample in front, optional two-source channel
// concat) as an MFMA implicit GEMM for gfx950.  Included by conv_halo.hip (two waves per SIMD) and conv_halo_big.hip
// (one wave per SIMD).
//
//   out[pixel][n] = epilogue( sum_{tap, c} X[pixel + tap][c] * W[n][tap][c] )
//
// Replaces the 3x3 nn.Conv2d of ResBlock.in_layers / out_layers, Upsample.conv and the VAE ResnetBlocks
// (/root/reference/lib/model_zoo/openaimodel.py:89-117,254-274, autokl_modules.py:82-141) on the shapes it accepts;
// everything else (stride 2, ragged grids, tiny Cout) stays on gemm_f16_kernel, whose gather re-fetches the activation
// tile for every one of the 9 taps (gemm_kernel.h: issue_begin).
//
// What is different from gemm_f16_kernel:
//   * a block owns BM output pixels that form an 8-row x 32-column patch of one image (or whole small images), and walks
//     the input channels in chunks of 64.  For a chunk the (rows + 2) x (cols + 2) pixel HALO of the patch is brought into
//     LDS ONCE by LDS-DMA ([halo pixel][64 channels], 128 bytes per pixel, 16-byte slots XOR-swizzled with the pixel index
//     on the source side); the 9 taps are 9 shifted views of that image: the fragment read of tap (ky, kx) is the same
//     ds_read_b128 at pixel + ky * pitch + kx.  Activation bytes through L2 -> LDS drop by 9 * BM / halo = 6.8x (256-pixel
//     patches: 340 halo pixels instead of 9 x 256); zero padding is the buffer descriptor's out-of-range read.
//   * only the weight tile [BN][64] of the current (chunk, tap) streams per K tile, so the DMA volume per MFMA is what a
//     (2.2 x larger) plain tile would need, and BM can be 256 pixels at BN = 160: N = 320 / 640 / 1280 are covered without
//     padding columns and a 64x64-latent conv is exactly one block per CU.
//   * the halo of the NEXT chunk arrives in 1-KiB pieces spread over the taps of the current one (double-buffered).
//   * the barrier of a tap sits in the MIDDLE of it, 3 weight stages, and operand fragments are requested one k-step ahead
//     with two named register sets, the request / MFMA order pinned with sched_barrier -- nothing a wave needs right after
//     the barrier depends on it (the second half of the tap reads a stage published by the previous barrier), so the matrix
//     pipe does not drain at tap boundaries.  (MODE 2 of rounds 3-5; the other barrier placements -- top of the tap, unpinned,
//     DMA requests spread over the MFMA gaps, one barrier per chunk -- lost in the forward and were removed in round 6.)
//   * split-K runs over channel chunks (fp32 slabs + splitk_reduce_kernel of gemm.hip).
// Epilogue (bias / activation / alpha in registers, tile staged through LDS, 16-byte row segments + per-image row vector +
// residual on the way out) follows gemm_f16_kernel; output rows are mapped from patch order back to pixel order.
#pragma once
#include "gemm_kernel.h"

namespace {

// compile-time loop: f(std::integral_constant<int, LO>{}) ... f(std::integral_constant<int, HI - 1>{})
template <int LO, int HI, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (LO < HI) {
        f(std::integral_constant<int, LO>{});
        static_for<LO + 1, HI>(f);
    }
}

struct ConvHaloArgs {
    GemmArgs g;                     // descriptor as validated / normalised by plan_gemm (gemm.hip) + operand extents
    int ltw, tw, rg, ngrp, lgsz;    // patch: tw columns (log2 ltw) x rg rows per group, ngrp groups (images), lgsz = log2(tw * rg)
    int pitch, gpx, hpx;            // halo: pixels per row (tw + 2), per group ((rg + 2) * pitch), in total
    int mg_pitch, mg_gpx;           // 2^20 / d + 1 magic numbers of the two divisions (operands < 2^20 / d)
    int tiles_x, tiles_y;           // patches per image (ngrp == 1)
    int nchunks, chunks_per_split;  // 64-channel chunks of the (concatenated) input
    int Hv, Wv;                     // (virtual, i.e. upsampled) image size == output size
    int halo_bytes;                 // one halo buffer: hpx rounded up to whole 8-pixel DMA pieces, x 128
    int abl;                        // -DVD_HALO_ABLATIONS builds only (timing experiments, wrong results): VD_HALO_ABL
    int nskip, skip_cps;            // SKIP instances: 64-channel chunks of the folded 1x1 skip convolution, per split
    unsigned s0_bytes, s1_bytes, sw_bytes;   // extents of the skip operands (buffer descriptors)
};
#ifdef VD_HALO_ABLATIONS
#define HALO_ABL(p, k) ((p).abl == (k))
#else
#define HALO_ABL(p, k) false
#endif

// MODE: 2 (the only main loop left, see above).  NT = 64 * waves; wave grid (BM / WM) x (BN / WN), wave tile WM pixels x WN channels.
// SKIP (round 4): behind the 3x3 chunks the block multiplies the chunks of a 1x1 convolution of a second (two-source) input on
// the same pixels into the same accumulators -- ResBlock's skip_connection(x) + h as extra K of the second conv (d.skip_*).
template <int BM, int BN, int WM, int WN, int NT, int MODE, bool SKIP = false>
__global__ __launch_bounds__(NT, NT / 256) void conv3x3_halo_kernel(const ConvHaloArgs p) {
    VD_TL_DECL;
    VD_TL(0);
    constexpr int NW = NT / 64;
    constexpr int WAVES_N = BN / WN, WAVES_M = BM / WM;
    static_assert(WAVES_M * WAVES_N == NW, "waves must tile the block");
    constexpr int MI = WM / 32, NI = WN / 32;
    static_assert(MODE == 2, "one main loop: mid-tap barrier, three weight stages, pinned request order");
    constexpr int WTILE = BN * 128;                     // bytes of one weight tile [BN][64]
    constexpr int WSTAGE = WTILE;
    constexpr int NPW = BN / 8;                         // 1-KiB DMA pieces (8 rows) of a weight tile
    constexpr int WPW = (NPW + NW - 1) / NW;            // ... per wave
    constexpr int HPXMAX = BM * 100 / 64 + 16;          // bound on halo pixels (checked by the launcher)
    constexpr int NHP = (HPXMAX + 7) / 8;               // halo pieces of 8 pixels
    constexpr int HPW = (NHP + NW - 1) / NW;            // ... per wave and chunk
    constexpr int HPT = (HPW + 7) / 8;                  // ... per wave and tap (taps 0..7 carry them)
    constexpr int MAXHP = HPT * 8;
    constexpr int CS_LD = BN + 8;                       // fp16 epilogue tile leading dimension
    static_assert(BN % 8 == 0 && WM % 32 == 0 && WN % 32 == 0, "tile shape");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const VdGemmDesc& d = p.g.d;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int hi = lane >> 5, l31 = lane & 31;

    // ---- XCD-aware tile mapping (as gemm_f16_kernel): an XCD gets a contiguous run of tiles, n fastest, so the column
    // tiles of one patch (same halo) and neighbouring patches (shared halo rows) meet in one L2
    const int ntiles = p.g.tiles_m * p.g.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = p.g.mfast ? bid % p.g.tiles_m : bid / p.g.tiles_n;   // see gemm_f16_kernel: which operand an XCD keeps
    const int tn = p.g.mfast ? bid / p.g.tiles_m : bid - tm * p.g.tiles_n;
    const int n0 = tn * BN;
    const int split = blockIdx.y;

    // patch origin
    int img0, y0, x0;
    if (p.ngrp == 1) {
        const int tpi = p.tiles_x * p.tiles_y;
        img0 = tm / tpi;
        const int r = tm - img0 * tpi;
        const int ty = r / p.tiles_x;
        y0 = ty * p.rg;
        x0 = (r - ty * p.tiles_x) * p.tw;
    } else {
        img0 = tm * p.ngrp;
        y0 = 0;
        x0 = 0;
    }

    const i32x4 ws_a0 = make_rsrc_words(d.a0, p.g.a0_bytes);
    const i32x4 ws_a1 = make_rsrc_words(d.a1 ? d.a1 : d.a0, d.a1 ? p.g.a1_bytes : 0u);
    const i32x4 ws_w = make_rsrc_words(d.w, p.g.w_bytes);

    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned w_lds0 = lds0 + 2u * (unsigned)p.halo_bytes;
    const char* const w_smem = smem + 2 * p.halo_bytes;

    // ---- per-lane source of the halo pieces this wave issues: piece q = j * NW + wave covers halo pixels 8q .. 8q + 7,
    // lane = (pixel in piece) * 8 + physical 16-byte slot; the lane fetches the LOGICAL slot that lives there (swizzle on
    // the source side, the DMA destination is lane-linear).  Packed as (input pixel index << 3 | logical slot), -1 = zeros.
    int hsrc[MAXHP];
#pragma unroll
    for (int j = 0; j < MAXHP; ++j) {
        const int hp = (j * NW + wave) * 8 + (lane >> 3);
        const int grp = (hp * p.mg_gpx) >> 20;
        const int rem = hp - grp * p.gpx;
        const int hy = (rem * p.mg_pitch) >> 20;
        const int hx = rem - hy * p.pitch;
        const int vy = y0 + hy - 1, vx = x0 + hx - 1;
        const bool ok = hp < p.hpx && (unsigned)vy < (unsigned)p.Hv && (unsigned)vx < (unsigned)p.Wv;
        const int pix = ((img0 + grp) * d.Hin + (vy >> d.ups)) * d.Win + (vx >> d.ups);
        const int slot = (lane & 7) ^ ((hp >> 1) & 7);
        hsrc[j] = ok ? ((pix << 3) | slot) : -1;
    }
    // weight pieces: piece q = j * NW + wave covers tile rows 8q .. 8q + 7
    unsigned wvoff[WPW];
#pragma unroll
    for (int j = 0; j < WPW; ++j) {
        const int r = (j * NW + wave) * 8 + (lane >> 3);
        const int n = n0 + r;
        const int slot = (lane & 7) ^ ((r >> 1) & 7);
        wvoff[j] = (r < BN && n < d.N) ? (unsigned)((n * d.ldw + slot * 8) * 2) : OOB_OFFSET;
    }

    const int ctot = d.c0 + d.c1;
    const int c_begin = split * p.chunks_per_split;
    int c_end = c_begin + p.chunks_per_split;
    if (c_end > p.nchunks) c_end = p.nchunks;
    const int ncl = HALO_ABL(p, 2) ? 0 : c_end - c_begin;   // >= 1 by construction of the launcher

    struct ChunkSrc { i32x4 rs; int ld2; unsigned soff; };
    auto chunk_src = [&](int c) {   // which tensor / channel offset a 64-channel chunk of the concatenation comes from
        ChunkSrc s;
        const int cc = c * 64;
        const bool second = cc >= d.c0;
        s.rs = second ? ws_a1 : ws_a0;
        s.ld2 = (second ? d.lda1 : d.lda0) * 2;
        s.soff = (unsigned)((second ? cc - d.c0 : cc) * 2);
        return s;
    };
    bool dma_on = true;
    auto issue_halo = [&](auto jt, const ChunkSrc& cs, unsigned buf_lds) {   // piece j of a chunk -> halo buffer at buf_lds
        constexpr int j = decltype(jt)::value;
        if (!dma_on) return;
        const int q = j * NW + wave_s;
        if (q * 8 < p.hpx) {   // wave-uniform
            const int h = hsrc[j];
            const unsigned voff = h < 0 ? OOB_OFFSET : (unsigned)((h >> 3) * cs.ld2 + ((h & 7) << 4));
            dma16(cs.rs, buf_lds + (unsigned)(q * 1024), voff, cs.soff);
        }
    };
    auto issue_w = [&](int c, int tap, int stage) {   // weight tile of (chunk c, tap) -> stage
        if (!dma_on) return;
        const unsigned soff = (unsigned)((tap * ctot + c * 64) * 2);
#pragma unroll
        for (int j = 0; j < WPW; ++j) {
            const int q = j * NW + wave_s;
            if (q < NPW) dma16(ws_w, w_lds0 + (unsigned)(stage * WSTAGE + q * 1024), wvoff[j], soff);
        }
    };

    // acc[i][j]: TRANSPOSED 32x32 sub-tile (MFMA A operand = weight rows, B operand = pixels): a lane owns output pixel
    // l31 of fragment i and, per register group g = r >> 2, channels 8g + 4hi + (r & 3) of fragment j.
    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // halo pixel of tap (0, 0) for each of the lane's output pixels
    int hp_base[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = wm * WM + i * 32 + l31;
        const int grp = m >> p.lgsz;
        const int r = m - (grp << p.lgsz);
        hp_base[i] = grp * p.gpx + (r >> p.ltw) * p.pitch + (r & (p.tw - 1));
    }
    // weight fragment offsets: swizzle key is the same for the NI fragments of a lane (32 rows apart)
    int rd_w[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) rd_w[ks] = lds_off_kb<64>(wn * WN + l31, ks * 2 + hi);

    // pixel fragments: byte = hp * 128 + (((2 ks + hi) ^ key) << 4), key = (hp >> 1) & 7, hp = halo pixel of the tap
    //                       = (hp * 128 + ((hi ^ key) << 4)) ^ (ks << 5): one address per (tap, fragment), one XOR per k-step.
    // The per-tap addresses are recomputed inside the chunk loop from `hrow` (made opaque per chunk): hoisted out of the
    // loop they cost 36 registers per fragment, which the operand double-buffering needs more.
    struct TapAddr { int a0[MI]; };
    int hrow[MI];
    auto tap_addr = [&](int halo_off, int tapoff) {   // halo_off: byte offset of the chunk's halo buffer in LDS
        TapAddr t;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int hp = hrow[i] + tapoff;
            const int x = ((hp >> 1) & 7) ^ hi;
            t.a0[i] = halo_off + (hp << 7) + (x << 4);
        }
        return t;
    };
    auto read_frags = [&](const char* wst, const TapAddr& t, int ks, f16x8* a, f16x8* w) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            U4H8 v;
            v.u = *reinterpret_cast<const uint4*>(smem + (t.a0[i] ^ (ks << 5)));
            a[i] = v.h;
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            U4H8 v;
            v.u = *reinterpret_cast<const uint4*>(wst + rd_w[ks] + j * 32 * 128);
            w[j] = v.h;
        }
    };
    auto mma = [&](const f16x8* a, const f16x8* w) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[j], a[i], acc[i][j], 0, 0, 0);
    };
    auto tapoff_of = [&](int t) { return (t / 3) * p.pitch + (t % 3); };

    // ---- prologue: the whole halo of the first chunk and the first weight tile(s)
    {
        const ChunkSrc cs0 = chunk_src(c_begin);
        static_for<0, MAXHP>([&](auto jt) { issue_halo(jt, cs0, lds0); });
        issue_w(c_begin, 0, 0);
        issue_w(c_begin, 1, 1);
    }

    if (HALO_ABL(p, 3)) dma_on = false;
    {
        // ---- barrier in the middle of a tap, fragments one k-step ahead (sets F0 / F1), weight tiles two taps ahead
        f16x8 a0f[MI], w0f[NI], a1f[MI], w1f[NI];
        wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#ifndef VD_TIMELINE_EPI
        VD_TL(1);   // first halo + weight tiles landed
#endif
#pragma unroll
        for (int i = 0; i < MI; ++i) hrow[i] = hp_base[i];
        {
            const TapAddr ta = tap_addr(0, 0);
            read_frags(w_smem, ta, 0, a0f, w0f);
        }
        for (int lc = 0; lc < ncl; ++lc) {
            const int c = c_begin + lc;
            const int halo_off = (lc & 1) * p.halo_bytes;
            const unsigned nxt_halo = lds0 + (unsigned)(((lc & 1) ^ 1) * p.halo_bytes);
            const bool more = lc + 1 < ncl;
            const ChunkSrc csn = chunk_src(more ? c + 1 : c);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                hrow[i] = hp_base[i];
                asm volatile("" : "+v"(hrow[i]));
            }
            static_for<0, 9>([&](auto tt) {
                constexpr int t = decltype(tt)::value;
                constexpr int stage = t % 3;   // (9 lc + t) % 3
                const char* wst = w_smem + stage * WSTAGE;
                const TapAddr ta = tap_addr(halo_off, tapoff_of(t));
                auto pin = [&]() { __builtin_amdgcn_sched_barrier(0); };
                read_frags(wst, ta, 1, a1f, w1f);
                pin();
                mma(a0f, w0f);
                pin();
                read_frags(wst, ta, 2, a0f, w0f);
                pin();
                mma(a1f, w1f);
                // the weight tile of the NEXT tap (issued one tap ago) has landed for every wave; every wave has left
                // the previous tap, whose stage the tile issued below overwrites
                wait_vm<0>();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if constexpr (t + 2 < 9) issue_w(c, t + 2, (t + 2) % 3);
                else if (more) issue_w(c + 1, t + 2 - 9, (t + 2) % 3);
                if constexpr (t < 8) {
                    if (more) static_for<t * HPT, (t + 1) * HPT>([&](auto jt) { issue_halo(jt, csn, nxt_halo); });
                }
                read_frags(wst, ta, 3, a1f, w1f);
                pin();
                mma(a0f, w0f);
                pin();
                if constexpr (t < 8) {
                    const TapAddr tn_ = tap_addr(halo_off, tapoff_of(t + 1));
                    read_frags(w_smem + ((t + 1) % 3) * WSTAGE, tn_, 0, a0f, w0f);
                } else if (more) {
                    const TapAddr tn_ = tap_addr(halo_off ^ p.halo_bytes, 0);
                    read_frags(w_smem, tn_, 0, a0f, w0f);
                }
                pin();
                mma(a1f, w1f);
            });
        }
    }
    if constexpr (SKIP) {
        // ---- folded 1x1 skip convolution: its 64-channel chunks are one-tap chunks (the centre pixel of the halo) with their
        // own weight matrix [N][skip_ldw]; this split takes its share of them.  The main pipeline is drained first, then a
        // plain two-buffer loop: wait + barrier, request the next chunk's halo and weight tile, four k-steps at the centre tap.
        // A chunk is 20 MFMAs per wave against 63 KB of DMA: the loop runs at the speed of the loads (~1.2 us per chunk), which
        // is what the separate GEMM's whole existence (launch, cold start, epilogue, the residual written and re-read) costs more.
        const int s_begin = split * p.skip_cps;
        int s_end = s_begin + p.skip_cps;
        if (s_end > p.nskip) s_end = p.nskip;
        if (s_begin < s_end) {
            wait_vm<0>();
            __builtin_amdgcn_s_barrier();   // main loop drained for every wave: halo buffers and weight stages are free
            asm volatile("" ::: "memory");
            const i32x4 ws_s0 = make_rsrc_words(d.skip_a0, p.s0_bytes);
            const i32x4 ws_s1 = make_rsrc_words(d.skip_a1 ? d.skip_a1 : d.skip_a0, d.skip_a1 ? p.s1_bytes : 0u);
            const i32x4 ws_sw = make_rsrc_words(d.skip_w, p.sw_bytes);
            unsigned wv2[WPW];
#pragma unroll
            for (int j = 0; j < WPW; ++j) {
                const int r = (j * NW + wave) * 8 + (lane >> 3);
                const int n = n0 + r;
                const int slot = (lane & 7) ^ ((r >> 1) & 7);
                wv2[j] = (r < BN && n < d.N) ? (unsigned)((n * d.skip_ldw + slot * 8) * 2) : OOB_OFFSET;
            }
            auto skip_src = [&](int c) {
                ChunkSrc s;
                const int cc = c * 64;
                const bool second = cc >= d.skip_c0;
                s.rs = second ? ws_s1 : ws_s0;
                s.ld2 = (second ? d.skip_lda1 : d.skip_lda0) * 2;
                s.soff = (unsigned)((second ? cc - d.skip_c0 : cc) * 2);
                return s;
            };
            auto issue_skip = [&](int c, int b) {   // whole halo of skip chunk c -> halo buffer b, its weight tile -> stage b
                const ChunkSrc cs = skip_src(c);
                static_for<0, MAXHP>([&](auto jt) { issue_halo(jt, cs, lds0 + (unsigned)(b * p.halo_bytes)); });
#pragma unroll
                for (int j = 0; j < WPW; ++j) {
                    const int q = j * NW + wave_s;
                    if (q < NPW) dma16(ws_sw, w_lds0 + (unsigned)(b * WSTAGE + q * 1024), wv2[j], (unsigned)(c * 64 * 2));
                }
            };
            issue_skip(s_begin, 0);
            for (int c = s_begin; c < s_end; ++c) {
                const int b = (c - s_begin) & 1;
                wait_vm<0>();
                __builtin_amdgcn_s_barrier();   // this chunk has landed for every wave; every wave has left the previous one
                asm volatile("" ::: "memory");
                if (c + 1 < s_end) issue_skip(c + 1, b ^ 1);
#pragma unroll
                for (int i = 0; i < MI; ++i) hrow[i] = hp_base[i];
                const TapAddr ta = tap_addr(b * p.halo_bytes, p.pitch + 1);   // centre tap (1, 1)
                const char* wst = w_smem + b * WSTAGE;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    f16x8 af[MI], wf[NI];
                    read_frags(wst, ta, ks, af, wf);
                    mma(af, wf);
                }
            }
        }
    }
    wait_vm<0>();
    __syncthreads();   // every wave is done with halo / weight stages: the epilogue tile re-uses that LDS
    VD_TL(2);

    const EpiCtx e = make_epi(d, 0);
    // patch-order row -> output row (pixel index over [image][Hv][Wv])
    auto out_row = [&](int m) {
        const int grp = m >> p.lgsz;
        const int r = m - (grp << p.lgsz);
        return ((img0 + grp) * p.Hv + y0 + (r >> p.ltw)) * p.Wv + x0 + (r & (p.tw - 1));
    };

    // ---- split over channel chunks: fp32 slabs for splitk_reduce_kernel, straight from registers
    if (gridDim.y > 1) {
        float* base = d.ws + (size_t)split * (size_t)d.M * d.N;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int row = out_row(wm * WM + i * 32 + l31);
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = n0 + wn * WN + j * 32 + 8 * g + 4 * hi;
                    if (col < d.N) {   // N % 8 == 0: whole groups
                        float* o = base + (size_t)row * d.N + col;
                        *reinterpret_cast<float4*>(o) = make_float4(acc[i][j][g * 4], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]);
                    }
                }
        }
        VD_TL(4);
        VD_TL_FLUSH(p.g.tl);
        return;
    }

    // ---- fused epilogue, part 1 (registers): bias -> act -> * alpha -> fp16 into the LDS tile [BM][CS_LD];
    // the residual OR row-vector segments of part 2 are requested first so their latency overlaps it
    f16* cs = reinterpret_cast<f16*>(smem);
    constexpr int CH = BN / 8;                    // 16-byte segments per tile row
    constexpr int MAX_CH = BM * CH / NT;
    static_assert(BM * CH % NT == 0, "segments must divide over the threads");
    const bool want_res = (e.flags & VD_EPI_RESIDUAL) != 0;
    const bool want_rv = (e.flags & VD_EPI_ROWVEC) != 0;
    const bool ld_ok = ((e.ldr & 7) == 0) && ((e.ldc & 7) == 0);
    const bool rv_per_image = e.rows_per_batch == p.Hv * p.Wv;
    // d.out_stats: per-channel statistics of the stored tile for a consuming GroupNorm (gemm_kernel.h: emit_chan_stats);
    // part 2 then writes what it stores back into the tile
    const bool want_stats = d.out_stats != nullptr && p.g.stat_rows > 0;
    auto rv_index = [&](int m, int row) { return rv_per_image ? img0 + (m >> p.lgsz) : row / e.rows_per_batch; };
    // few segments per thread: request them now; many (one wave per SIMD): read in line, the registers are not there
    constexpr bool PREFETCH = MAX_CH <= 12;
    constexpr int NPRE = PREFETCH ? MAX_CH : 1;
    uint4 pre[NPRE];
    // Every operand of the epilogue is requested up front and BRANCH-FREE (buffer loads: an out-of-range offset returns zeros),
    // so the requests leave back to back.  Rounds 1-3 wrote `if (col < N) pre[k] = *ptr` / `if (bias) t = *ptr`: hipcc put each
    // load into its own conditional block with s_waitcnt vmcnt(0) behind it -- 10 + 20 SERIAL round trips per block, 11.8 us
    // (18.6 us with a cold residual) of the 72-us life of a 64x64-level block (tools/probes/gemm_timeline.py).
    if constexpr (PREFETCH) {
        const __amdgpu_buffer_rsrc_t rs_pre = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<f16*>(want_res ? e.res : e.rowvec), 0, (ld_ok && (want_res || want_rv) && !HALO_ABL(p, 1)) ? 0x7fffffff : 0, 0x00020000);
#pragma unroll
        for (int k = 0; k < MAX_CH; ++k) {
            const int sgm = tid + k * NT;
            const int r = sgm / CH, col = n0 + (sgm % CH) * 8;
            const int row = out_row(r);
            const unsigned off = want_res ? (unsigned)((row * e.ldr + col) * 2) : (unsigned)((rv_index(r, row) * e.N + col) * 2);
            const vd_u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs_pre, (int)(col < e.N ? off : OOB_OFFSET), 0, 0);
            pre[k] = make_uint4(v[0], v[1], v[2], v[3]);
        }
    }
    U2H4 bias_r[NI * 4];
    {
        EpiCtx eb = e;
        eb.flags &= ~VD_EPI_BIAS_ALONG_M;   // (this kernel has no bias along M)
        epi_load_bias<NI>(eb, d.N, n0 + wn * WN + 4 * hi, bias_r, true);   // N % 8 == 0 here: whole groups
    }
#if defined(VD_TIMELINE) && defined(VD_TIMELINE_EPI)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    VD_TL(1);   // (-DVD_TIMELINE_EPI: slot 1 = the epilogue's operands have arrived)
#endif
    // (the activation bodies are compact -- gemm_kernel.h: apply_act -- so the ACT_NONE path of a group hops over ~40
    // instructions, not over ~400: with the three inlined IEEE-division bodies this loop took 3.8 us for 80 values per lane,
    // instruction fetch, not arithmetic)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int lrow = wm * WM + i * 32 + l31;
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int lc = wn * WN + j * 32 + 8 * g + 4 * hi;
                float bq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) bq[q] = (float)bias_r[j * 4 + g].e[q];   // zeros without a bias / past N
                U2H4 o;
                if (e.act == VD_ACT_NONE) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) o.e[q] = (f16)((acc[i][j][g * 4 + q] + bq[q]) * e.alpha);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) o.e[q] = (f16)(apply_act(e.act, acc[i][j][g * 4 + q] + bq[q]) * e.alpha);
                }
                *reinterpret_cast<uint2*>(cs + lrow * CS_LD + lc) = o.u;
            }
    }
    __syncthreads();
    VD_TL(3);

    // ---- part 2: 16-byte row segments: (+ rowvec) (+ residual) -> global
    if (PREFETCH && ld_ok && !(want_res && want_rv) && !HALO_ABL(p, 1)) {
        auto part2 = [&](auto nt_tag, auto keep_tag) {
            constexpr bool NTS = decltype(nt_tag)::value, KEEP = decltype(keep_tag)::value;
#pragma unroll
            for (int k = 0; k < MAX_CH; ++k) {
                const int sgm = tid + k * NT;
                const int r = sgm / CH, cc = (sgm % CH) * 8;
                const int col = n0 + cc;
                if (col < e.N) {
                    const int row = out_row(r);
                    U4H8 t, a, o;
                    t.u = *reinterpret_cast<const uint4*>(cs + r * CS_LD + cc);
                    a.u = pre[PREFETCH ? k : 0];
#pragma unroll
                    for (int q = 0; q < 8; ++q) o.e[q] = (f16)((float)t.e[q] + (float)a.e[q]);
                    f16* dst = reinterpret_cast<f16*>(e.out) + (size_t)row * e.ldc + col;
                    if constexpr (NTS) vd_store16_nt(dst, o.u);
                    else *reinterpret_cast<uint4*>(dst) = o.u;
                    if constexpr (KEEP) *reinterpret_cast<uint4*>(cs + r * CS_LD + cc) = o.u;
                }
            }
        };
        if (p.g.nt_store) {
            if (want_stats) part2(std::true_type{}, std::true_type{});
            else part2(std::true_type{}, std::false_type{});
        } else {
            if (want_stats) part2(std::false_type{}, std::true_type{});
            else part2(std::false_type{}, std::false_type{});
        }
    } else {
#pragma unroll
    for (int k = 0; k < MAX_CH; ++k) {
        const int sgm = tid + k * NT;
        const int r = sgm / CH, cc = (sgm % CH) * 8;
        const int col = n0 + cc;
        if (col < e.N && !(HALO_ABL(p, 1) && r + cc != -1)) {
            const int row = out_row(r);
            U4H8 t, a, b, o;
            t.u = *reinterpret_cast<const uint4*>(cs + r * CS_LD + cc);
            if (ld_ok) {
                a.u = make_uint4(0, 0, 0, 0);
                b.u = make_uint4(0, 0, 0, 0);
                if constexpr (PREFETCH) {
                    a.u = pre[k];
                    if (want_res && want_rv) b.u = *reinterpret_cast<const uint4*>(e.rowvec + (size_t)rv_index(r, row) * e.N + col);
                } else {
                    if (want_res) a.u = *reinterpret_cast<const uint4*>(e.res + (size_t)row * e.ldr + col);
                    if (want_rv) b.u = *reinterpret_cast<const uint4*>(e.rowvec + (size_t)rv_index(r, row) * e.N + col);
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) o.e[q] = (f16)((float)t.e[q] + (float)a.e[q] + (float)b.e[q]);
                f16* dst = reinterpret_cast<f16*>(e.out) + (size_t)row * e.ldc + col;
                if (p.g.nt_store) vd_store16_nt(dst, o.u);
                else *reinterpret_cast<uint4*>(dst) = o.u;
                if (want_stats) *reinterpret_cast<uint4*>(cs + r * CS_LD + cc) = o.u;
            } else {   // unaligned leading dimensions: element-wise tail of gemm_f16_kernel
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = (float)t.e[q];
                epi_finish8(e, row, col, v);
            }
        }
    }
    }
    VD_TL(5);   // part 2 stores issued
    // one partial per patch (a patch lies in one image), or per image where a patch holds several whole small images
    if (want_stats) {
        __syncthreads();
        const int nsub = p.ngrp, R = BM / p.ngrp;
        emit_chan_stats<BN, CS_LD, NT>(cs, reinterpret_cast<float*>(smem + BM * CS_LD * 2), tid, R, nsub, nsub, d.out_stats,
                                       (size_t)tm * nsub, d.N, n0, reinterpret_cast<unsigned long long*>(d.stat_sums), d.stat_img_rows);
    }
    VD_TL(4);   // (behind the statistics pass where one runs)
    VD_TL_FLUSH(p.g.tl);
}

template <int BM, int BN, int WM, int WN, int NT, int MODE, bool SKIP = false>
int launch_conv_halo(const ConvHaloArgs& a, int nsplit, hipStream_t stream) {
    constexpr int WST = 3;
    constexpr int EPI = stat_lds_bytes(BM, BN);   // epilogue tile + the lane scratch of the statistics pass
    const int main_bytes = 2 * a.halo_bytes + WST * BN * 128;
    const int lds = main_bytes > EPI ? main_bytes : EPI;
    if (lds > 160 * 1024) {
        vd_set_error("conv3x3_halo: %d bytes of LDS", lds);
        return VD_ERR_UNSUPPORTED;
    }
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<BM, BN, WM, WN, NT, MODE, SKIP>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) {
            vd_set_error("conv3x3_halo: cannot reserve LDS: %s", hipGetErrorString(e));
            return VD_ERR_LAUNCH;
        }
        done.fetch_or(bit, std::memory_order_release);
    }
    dim3 grid(a.g.tiles_m * a.g.tiles_n, nsplit, 1);
    hipLaunchKernelGGL((conv3x3_halo_kernel<BM, BN, WM, WN, NT, MODE, SKIP>), grid, dim3(NT), lds, stream, a);
    return vd_check_launch("conv3x3_halo");
}

}  // namespace
