// Weight-streaming 3x3 convolution for the 8x8 level of the UNet (round 4): M = images x 64 pixels is tiny (512 rows at the
// bench shape), N x K = 1280 x 11520 .. 23040 -- the layer IS its weight stream (29.5 / 59 MB against 1.3 / 2.6 MB of
// activations).  Included by gemm.hip (vd_conv3x3_wstream_f16).
//
//   out[pixel][n] = epilogue( sum_{tap, c} X[pixel + tap][c] * W[n][tap][c] ),   stride 1, pad 1, 8 x 8 images
//
// Replaces the 3x3 nn.Conv2d of the ResBlocks at ds = 8 (input blocks 10 / 11, the middle block, output blocks 0-2:
// /root/reference/lib/model_zoo/openaimodel.py:254-274) where rounds 1-3 ran gemm_f16_kernel with a 9-12-way split over K:
// 128 x 64 tiles re-fetched activations for every tap and every weight tile four times (350 MB through the L2 -> LDS path for
// 15 GFLOP: 41-47 us + a reduce launch, 0.15 of the MFMA peak).
//
// What is different here:
//   * weights never touch LDS.  The host packs them once in MFMA-fragment order ([n tile of 32][64-channel chunk][tap][k-step]
//     [lane][8 halfs], vd_hip/pack.py: pack_conv_weight_stream): the A operand of one v_mfma_f32_32x32x16_f16 is ONE fully
//     coalesced 1-KiB global load per wave, used for every pixel tile the wave owns and then dropped.  Each weight byte enters
//     exactly one wave per image group; a ring of D k-steps of fragments in registers covers the HBM latency (plain loads: the
//     compiler counts vmcnt).
//   * a block = 4 waves x 64 output channels (2 n tiles each) over the SAME 128 pixels (2 images, 4 pixel tiles): 8 MFMAs per
//     k-step and wave from 2 weight fragments (registers) + 4 pixel fragments (LDS) -- 0.5 ds_read_b128 per MFMA.
//   * the 2 x (10 x 10) pixel halo of a 64-channel chunk is brought to LDS once by LDS-DMA (28 KiB, double-buffered, one
//     barrier per chunk); the nine taps read shifted views (as conv3x3_halo_kernel).  Halo rows have an ODD pitch (11 pixels)
//     and the 16-byte slots of a pixel are XOR-swizzled with its halo COLUMN, so the 16 lanes of a ds_read_b128 group -- two
//     image rows of 8 pixels -- always hit 16 distinct bank quads, for every tap.
//   * K is split over 64-channel chunks across blocks (fp32 slabs + the split-K reduce kernels of gemm.hip, which run the fused
//     epilogue and, on request, emit the GroupNorm statistics): ~10 splits x 20 tiles = 200 blocks, two per CU co-resident.
#pragma once
#include "gemm_kernel.h"

namespace {

struct WsArgs {
    const f16* a0; const f16* a1; const uint4* wp; float* ws;
    int c0, c1, lda0, lda1;
    int nimg, M, N;
    int nchunks, cps;          // 64-channel chunks of the (concatenated) input: in total / per split
    int tiles_m, tiles_n, nsplit;
    unsigned a0_bytes, a1_bytes;
    // folded 1x1 skip convolution (ResBlock skip_connection as extra K): one-tap chunks behind the 3x3 chunks
    const f16* s0; const f16* s1; const uint4* swp;   // sources on the same 8x8 grid; weights [N / 32][chunks][4 k-steps][64 lanes] x 16 B
    int sc0, sc1, slda0, slda1, nskip, skip_cps;
    unsigned s0_bytes, s1_bytes;
};

constexpr int WS_PITCH = 11, WS_GPX = 110, WS_IPB = 2, WS_NPIECE = 28, WS_HB = WS_NPIECE * 1024;

template <int LO, int HI, class F>
__device__ __forceinline__ void ws_static_for(F&& f) {
    if constexpr (LO < HI) {
        f(std::integral_constant<int, LO>{});
        ws_static_for<LO + 1, HI>(f);
    }
}

// D = k-steps of weight fragments in flight per wave (divides 36); OCC = waves per SIMD the register budget is set for;
// IPB = images per block.  IPB = 2 (round 4): 128 pixels, a wave owns 64 output channels (2 n tiles) -> 256 channels per block.
// IPB = 4 (round 5): 256 pixels, a wave owns 32 output channels (1 n tile) over all eight pixel tiles -> 128 channels per block:
// the same 8 MFMAs per k-step and wave, but every weight byte now enters HALF as many CUs.  The launch is bound by what a CU
// ingests through its L1 (~27 GB/s per CU for global -> register loads, measured on the whole-K variant: conv_wsk_kernel.h), and
// the bench shape's 29.5 MB of weights used to enter 4 CUs each (one per image pair).
template <int D, int OCC, int IPB = 2>
__global__ __launch_bounds__(256, OCC) void conv3x3_wstream_kernel(const WsArgs p) {
    static_assert(36 % D == 0, "the fragment ring must divide the 36 k-steps of a chunk");
    static_assert(IPB == 2 || IPB == 4, "image groups of 2 or 4");
    constexpr int NPX = IPB * 2;                        // 32-pixel tiles of the block (all of them in every wave)
    constexpr int NTW = 4 / IPB;                        // n tiles (32 channels) per wave
    constexpr int NPIECE = (IPB * WS_GPX * 128 + 1023) / 1024;
    constexpr int PPW = (NPIECE + 3) / 4;               // halo pieces per wave and chunk
    constexpr int HB = 4 * PPW * 1024;                  // bytes of a halo buffer
    static_assert((36 - PPW) * NTW >= NTW * D, "the halo pieces of a chunk must be older than the fragment ring");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int hi = lane >> 5, l31 = lane & 31;

    // blocks that share a weight panel (same column slice and K split, the tiles_m image groups) get consecutive logical
    // indices inside one XCD's contiguous run: the panel is fetched from HBM once and hits that L2 for the other groups
    const int ntot = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = ntot >> 3, r = ntot & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid % p.tiles_m;
    const int rest = bid / p.tiles_m;
    const int tn = rest % p.tiles_n;
    const int split = rest / p.tiles_n;
    const int img0 = tm * IPB;
    const int n0 = tn * (128 * NTW) + wave_s * (32 * NTW);
    const int c_begin = split * p.cps;
    int c_end = c_begin + p.cps;
    if (c_end > p.nchunks) c_end = p.nchunks;
    const int ncl = c_end - c_begin;   // >= 1 by construction of the launcher

    const i32x4 rs_a0 = make_rsrc_words(p.a0, p.a0_bytes);
    const i32x4 rs_a1 = make_rsrc_words(p.a1 ? p.a1 : p.a0, p.a1 ? p.a1_bytes : 0u);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // ---- halo pieces of this wave: piece q = j * 4 + wave covers halo pixels 8q .. 8q + 7, lane = (pixel in piece) * 8 +
    // physical slot; packed (input pixel << 3 | logical slot), -1 = zeros (padding ring, pitch filler, tail)
    int hsrc[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int hp = (j * 4 + wave) * 8 + (lane >> 3);
        const int g = hp / WS_GPX;
        const int rem = hp - g * WS_GPX;
        const int yh = (rem * 373) >> 12;        // rem / 11 for rem < 128
        const int xh = rem - yh * WS_PITCH;
        const bool ok = hp < IPB * WS_GPX && yh >= 1 && yh <= 8 && xh >= 1 && xh <= 8;
        const int pix = ((img0 + g) * 8 + yh - 1) * 8 + xh - 1;
        hsrc[j] = ok ? ((pix << 3) | ((lane & 7) ^ (xh & 7))) : -1;
    }
    struct ChunkSrc { i32x4 rs; int ld2; unsigned soff; };
    auto chunk_src = [&](int c) {
        ChunkSrc s;
        const int cc = c * 64;
        const bool second = cc >= p.c0;
        s.rs = second ? rs_a1 : rs_a0;
        s.ld2 = (second ? p.lda1 : p.lda0) * 2;
        s.soff = (unsigned)((second ? cc - p.c0 : cc) * 2);
        return s;
    };
    auto issue_halo = [&](auto jt, const ChunkSrc& cs, unsigned buf_lds) {
        constexpr int j = decltype(jt)::value;
        const int h = hsrc[j];
        const unsigned voff = h < 0 ? OOB_OFFSET : (unsigned)((h >> 3) * cs.ld2 + ((h & 7) << 4));
        dma16(cs.rs, buf_lds + (unsigned)((j * 4 + wave_s) * 1024), voff, cs.soff);
    };

    // ---- weight stream of this wave: NTW n tiles, 36 fragments (1 KiB each) per chunk and tile, consecutive chunks contiguous
    const int nt0 = n0 >> 5;
    const uint4* wq0 = p.wp + ((size_t)nt0 * p.nchunks + c_begin) * (36 * 64) + lane;
    const uint4* wq1 = p.wp + ((size_t)(nt0 + (NTW - 1)) * p.nchunks + c_begin) * (36 * 64) + lane;
    const int kmax = ncl * 36 - 1;
    U4H8 wf[D][NTW];
    auto load_w = [&](auto rt, int kk) {   // fragments of k-step kk (clamped: the tail re-reads the last one) -> ring slot
        constexpr int r = decltype(rt)::value;
        const int k = kk < kmax ? kk : kmax;
        wf[r][0].u = wq0[(size_t)k * 64];
        if constexpr (NTW == 2) wf[r][1].u = wq1[(size_t)k * 64];
    };

    f32x16 acc[NPX][NTW];
#pragma unroll
    for (int i = 0; i < NPX; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // pixel fragments: lane pixel m = i * 32 + l31 of the block's 128 -> image g = m >> 6, (y, x) = ((m >> 3) & 7, m & 7); halo
    // pixel of tap (ky, kx) = g * 110 + (y + ky) * 11 + x + kx, slot key = (x + kx) & 7:
    //   byte = pixel * 128 + ((((2 ks + hi) ^ key)) << 4) = hp0b[i] + (tkx[kx] ^ (ks << 5)) + (ky * 11 + kx) * 128
    int hp0b[NPX], tkx[3];   // (m & 7) == (l31 & 7) for every pixel tile: the slot term depends on kx only
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
        const int m = i * 32 + l31;
        hp0b[i] = ((m >> 6) * WS_GPX + ((m >> 3) & 7) * WS_PITCH + (m & 7)) * 128;
    }
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) tkx[kx] = ((((l31 & 7) + kx) & 7) ^ hi) << 4;
    auto read_b = [&](auto st, int buf_off, f16x8* b) {   // pixel fragments of k-step s = tap * 4 + ks of the chunk in buf_off
        constexpr int s = decltype(st)::value;
        constexpr int tap = s >> 2, ks = s & 3, ky = tap / 3, kx = tap % 3;
#pragma unroll
        for (int i = 0; i < NPX; ++i) {
            U4H8 v;
            v.u = *reinterpret_cast<const uint4*>(smem + buf_off + hp0b[i] + (tkx[kx] ^ (ks << 5)) + (ky * WS_PITCH + kx) * 128);
            b[i] = v.h;
        }
    };

    // ---- prologue: halo of the first chunk, the first D k-steps of weights
    {
        const ChunkSrc cs0 = chunk_src(c_begin);
        ws_static_for<0, PPW>([&](auto jt) { issue_halo(jt, cs0, lds0); });
    }
    ws_static_for<0, D>([&](auto rt) { load_w(rt, decltype(rt)::value); });

    f16x8 bf[2][NPX];
    for (int lc = 0; lc < ncl; ++lc) {
        const int buf_off = (lc & 1) * HB;
        const unsigned nxt_lds = lds0 + (unsigned)(((lc & 1) ^ 1) * HB);
        const bool more = lc + 1 < ncl;
        const ChunkSrc csn = chunk_src(more ? c_begin + lc + 1 : c_begin + lc);
        // this chunk's halo pieces (issued during the previous chunk, older than all but the youngest 2 * D weight loads) have
        // landed for this wave ... for every wave; every wave has left the previous chunk, whose buffer the DMA below refills
        wait_vm<NTW * D>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        read_b(std::integral_constant<int, 0>{}, buf_off, bf[0]);
        const int kk0 = lc * 36;
        ws_static_for<0, 36>([&](auto st) {
            constexpr int s = decltype(st)::value;
            if constexpr (s < 35) read_b(std::integral_constant<int, s + 1>{}, buf_off, bf[(s + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NPX; ++i)
#pragma unroll
                for (int j = 0; j < NTW; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s % D][j].h, bf[s & 1][i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_w(std::integral_constant<int, s % D>{}, kk0 + s + D);
            if constexpr (s < PPW) {
                if (more) issue_halo(std::integral_constant<int, s>{}, csn, nxt_lds);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    }

    // ---- folded 1x1 skip convolution: this split's share of its 64-channel chunks, 4 k-steps each at the centre tap.  The
    // work is tiny (32 MFMAs per chunk and wave): a plain loop -- request the chunk's halo and its 8 weight fragments, wait,
    // barrier, multiply -- instead of the ring.
    if (p.nskip > 0) {
        const int s_begin = split * p.skip_cps;
        int s_end = s_begin + p.skip_cps;
        if (s_end > p.nskip) s_end = p.nskip;
        const i32x4 rs_s0 = make_rsrc_words(p.s0, p.s0_bytes);
        const i32x4 rs_s1 = make_rsrc_words(p.s1 ? p.s1 : p.s0, p.s1 ? p.s1_bytes : 0u);
        const uint4* sq0 = p.swp + (size_t)nt0 * p.nskip * (4 * 64) + lane;
        const uint4* sq1 = p.swp + (size_t)(nt0 + (NTW - 1)) * p.nskip * (4 * 64) + lane;
        for (int c = s_begin; c < s_end; ++c) {
            wait_vm<0>();
            __builtin_amdgcn_s_barrier();   // every wave has left the buffer the DMA below refills (buffer 0 is re-used)
            asm volatile("" ::: "memory");
            ChunkSrc cs;
            {
                const int cc = c * 64;
                const bool second = cc >= p.sc0;
                cs.rs = second ? rs_s1 : rs_s0;
                cs.ld2 = (second ? p.slda1 : p.slda0) * 2;
                cs.soff = (unsigned)((second ? cc - p.sc0 : cc) * 2);
            }
            ws_static_for<0, PPW>([&](auto jt) { issue_halo(jt, cs, lds0); });
            U4H8 sw[4][NTW];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                sw[ks][0].u = sq0[((size_t)c * 4 + ks) * 64];
                if constexpr (NTW == 2) sw[ks][1].u = sq1[((size_t)c * 4 + ks) * 64];
            }
            wait_vm<0>();
            __builtin_amdgcn_s_barrier();   // the halo has landed for every wave
            asm volatile("" ::: "memory");
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                f16x8 b4[NPX];
#pragma unroll
                for (int i = 0; i < NPX; ++i) {
                    U4H8 v;
                    v.u = *reinterpret_cast<const uint4*>(smem + hp0b[i] + (tkx[1] ^ (ks << 5)) + (WS_PITCH + 1) * 128);   // tap (1, 1)
                    b4[i] = v.h;
                }
#pragma unroll
                for (int i = 0; i < NPX; ++i)
#pragma unroll
                    for (int j = 0; j < NTW; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sw[ks][j].h, b4[i], acc[i][j], 0, 0, 0);
            }
        }
    }

    // ---- fp32 slab of this split for the reduce kernel, straight from registers (4 consecutive floats per lane and group)
    float* base = p.ws + (size_t)split * (size_t)p.M * p.N;
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
        const int row = img0 * 64 + i * 32 + l31;
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = n0 + j * 32 + 8 * g + 4 * hi;
                *reinterpret_cast<float4*>(base + (size_t)row * p.N + col) =
                    make_float4(acc[i][j][g * 4], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]);
            }
    }
}

}  // namespace
