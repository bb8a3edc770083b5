// 3x3 convolution (stride 1, pad 1, optional nearest-2x upsample in front, optional two-source channel concat) with the
// WEIGHTS HELD IN REGISTERS: conv3x3_wstream_kernel's main loop (conv_wstream_kernel.h) generalised from the 8x8 level to
// every 3x3 convolution of the UNet / VAE whose output grid tiles into 128-pixel patches.  Included by conv_wstream.hip.
//
//   out[pixel][n] = epilogue( sum_{tap, c} X[pixel + tap][c] * W[n][tap][c] )
//
// Replaces the same reference lines as conv3x3_halo_kernel (ResBlock.in_layers / out_layers, Upsample.conv, the VAE
// ResnetBlocks: /root/reference/lib/model_zoo/openaimodel.py:89-117,254-274, autokl_modules.py:82-141).
//
// Why a second formulation next to the halo kernel.  There the weight tile of every (chunk, tap) goes global -> LDS by DMA
// and both MFMA operands are read from LDS (1.2 ds_read_b128 per MFMA at 32 x 160 wave tiles, a barrier per tap, three
// weight stages): 42-46 % of the MFMA peak on the largest layers, the waves parked at barriers / waitcnt for a third of their
// cycles.  Here (measured on conv3x3_wstream_kernel, tools/wstream_bench.py: 4.8 us per 64-channel chunk of 288 MFMAs per
// wave = the issue rate of the matrix pipe):
//   * the A operand (weights) of an MFMA is ONE coalesced 1-KiB global load per wave straight into registers from the
//     fragment-ordered copy of the weights (pack_conv_weight_stream); a ring of D k-steps in flight hides the latency; no
//     weight ever touches LDS, no barrier guards it;
//   * a wave owns 128 pixels (4 pixel tiles) x NI x 32 output channels (NI = 1 .. 4: 64 NI accumulator registers in AGPRs, one
//     wave per SIMD): a weight fragment feeds 4 MFMAs (ingest 32 B / clk / CU), a pixel fragment NI of them (1 / NI
//     ds_read_b128 per MFMA);
//   * the four waves of a block share the 128-pixel patch (8 x 16 or 4 x 32 pixels of one image) and split the block's output
//     channels: (NI of waves 0-1, NI of waves 2-3) = (4,4) 512 channels, (3,3) 384, (3,2) 320, (2,2) 256, (1,1) 128; the host
//     tiles N with these (320 = one (3,2) block, 640 = (3,3) + (2,2), 1280 = 2 x (4,4) + (2,2));
//   * the (rows + 2) x (cols + 2) halo of a 64-channel chunk is staged in LDS once by LDS-DMA, double-buffered, ONE barrier
//     per chunk (layout and swizzle as conv3x3_halo_kernel: [halo pixel][64 ch], slot ^ ((pixel >> 1) & 7));
//   * no split: fused epilogue as conv3x3_halo_kernel (bias, activation, alpha in registers -> fp16 tile in LDS -> 16-byte row
//     segments + per-image row vector + residual, per-channel GroupNorm statistics on request); split over chunks: fp32 slabs
//     for the reduce kernels of gemm.hip.
#pragma once
#include "gemm_kernel.h"

namespace {

constexpr int WR_MAX_TN = 8;

struct WrArgs {
    GemmArgs g;                    // normalised descriptor (operands, epilogue), operand extents, nt_store, stat_rows
    const uint4* wp;               // weights in fragment order [N / 32][chunks][9][4][64 lanes] x 16 bytes
    int ltw, tw, rows, pitch, hpx; // patch: tw columns (log2 ltw) x rows = 128 / tw pixels; halo pitch tw + 2, (rows + 2) * pitch pixels
    int mg_pitch;                  // 2^20 / pitch + 1
    int tiles_x, tiles_y;          // patches per image
    int Hv, Wv;                    // (virtual, i.e. upsampled) image size == output size
    int nchunks, cps, nsplit;      // 64-channel chunks of the (concatenated) input: in total / per split / splits
    int tiles_m, ntn;              // patches, column tiles
    int tn_n0[WR_MAX_TN], tn_ni0[WR_MAX_TN], tn_ni1[WR_MAX_TN];   // first channel and wave layout of each column tile
    int halo_bytes;                // one halo buffer: hpx rounded up to whole 8-pixel pieces, x 128
};

template <int LO, int HI, class F>
__device__ __forceinline__ void wr_static_for(F&& f) {
    if constexpr (LO < HI) {
        f(std::integral_constant<int, LO>{});
        wr_static_for<LO + 1, HI>(f);
    }
}

// what every wave of the block needs, computed once
struct WrCtx {
    int tid, lane, wave, hi, l31;
    int tm, split, img0, y0, x0;
    int c_begin, ncl;
    int n0_blk, bn_blk;            // first channel / width of the block's column tile
    i32x4 rs_a0, rs_a1;
    unsigned lds0;
    int hsrc[7];
};

constexpr int wr_ring_depth(int NI) { return NI <= 2 ? 12 : (NI == 3 ? 9 : 6); }   // k-steps of weight fragments in flight

// one wave: NI n tiles starting at channel n_w (column cw inside the block's tile), all chunks of the split, then either the
// fp32 slab of the split or part 1 of the fused epilogue (accumulators -> fp16 tile in LDS)
template <int NI>
__device__ __forceinline__ void wr_wave(const WrArgs& p, const WrCtx& c, char* smem, const int n_w, const int cw) {
    constexpr int D = wr_ring_depth(NI);
    static_assert(36 % D == 0, "the fragment ring must divide the 36 k-steps of a chunk");
    const VdGemmDesc& d = p.g.d;
    const int lane = c.lane, hi = c.hi, l31 = c.l31;
    const int wave_s = __builtin_amdgcn_readfirstlane(c.wave);

    struct ChunkSrc { i32x4 rs; int ld2; unsigned soff; };
    auto chunk_src = [&](int ch) {
        ChunkSrc s;
        const int cc = ch * 64;
        const bool second = cc >= d.c0;
        s.rs = second ? c.rs_a1 : c.rs_a0;
        s.ld2 = (second ? d.lda1 : d.lda0) * 2;
        s.soff = (unsigned)((second ? cc - d.c0 : cc) * 2);
        return s;
    };
    auto issue_halo = [&](auto jt, const ChunkSrc& cs, unsigned buf_lds) {
        constexpr int j = decltype(jt)::value;
        const int q = j * 4 + wave_s;
        if (q * 8 < p.hpx) {   // wave-uniform
            const int h = c.hsrc[j];
            const unsigned voff = h < 0 ? OOB_OFFSET : (unsigned)((h >> 3) * cs.ld2 + ((h & 7) << 4));
            dma16(cs.rs, buf_lds + (unsigned)(q * 1024), voff, cs.soff);
        }
    };

    // weight stream: NI n tiles, 36 fragments (1 KiB) per chunk and tile, consecutive chunks contiguous
    const size_t tile_stride = (size_t)p.nchunks * (36 * 64);
    const uint4* wq = p.wp + ((size_t)(n_w >> 5) * p.nchunks + c.c_begin) * (36 * 64) + lane;
    const int kmax = c.ncl * 36 - 1;
    U4H8 wf[D][NI];
    auto load_w = [&](auto rt, int kk) {
        constexpr int r = decltype(rt)::value;
        const int k = kk < kmax ? kk : kmax;
#pragma unroll
        for (int j = 0; j < NI; ++j) wf[r][j].u = wq[(size_t)j * tile_stride + (size_t)k * 64];
    };

    f32x16 acc[4][NI];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // halo pixel of tap (0, 0) for each of the lane's 4 output pixels (pixel m = i * 32 + l31 of the patch)
    int hp_base[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = i * 32 + l31;
        hp_base[i] = (m >> p.ltw) * p.pitch + (m & (p.tw - 1));
    }
    struct TapAddr { int a0[4]; };
    auto tap_addr = [&](int halo_off, int tapoff) {
        TapAddr t;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int hp = hp_base[i] + tapoff;
            t.a0[i] = halo_off + (hp << 7) + (((((hp >> 1) & 7)) ^ hi) << 4);
        }
        return t;
    };
    auto read_b = [&](const TapAddr& t, int ks, f16x8* b) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            U4H8 v;
            v.u = *reinterpret_cast<const uint4*>(smem + (t.a0[i] ^ (ks << 5)));
            b[i] = v.h;
        }
    };

    // ---- prologue: halo of the first chunk, the first D k-steps of weights
    {
        const ChunkSrc cs0 = chunk_src(c.c_begin);
        wr_static_for<0, 7>([&](auto jt) { issue_halo(jt, cs0, c.lds0); });
    }
    wr_static_for<0, D>([&](auto rt) { load_w(rt, decltype(rt)::value); });

    f16x8 bf[2][4];
    for (int lc = 0; lc < c.ncl; ++lc) {
        const int buf_off = (lc & 1) * p.halo_bytes;
        const unsigned nxt_lds = c.lds0 + (unsigned)(((lc & 1) ^ 1) * p.halo_bytes);
        const bool more = lc + 1 < c.ncl;
        const ChunkSrc csn = chunk_src(more ? c.c_begin + lc + 1 : c.c_begin + lc);
        // this chunk's halo pieces (older than all but the youngest D * NI weight loads) have landed for this wave ... for every
        // wave; every wave has left the previous chunk, whose buffer the DMA below refills
        wait_vm<D * NI>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        TapAddr ta = tap_addr(buf_off, 0);
        read_b(ta, 0, bf[0]);
        const int kk0 = lc * 36;
        wr_static_for<0, 36>([&](auto st) {
            constexpr int s = decltype(st)::value;
            constexpr int tap = s >> 2, ks = s & 3;
            if constexpr (s < 35) {
                constexpr int tn_ = (s + 1) >> 2, kn = (s + 1) & 3;
                if constexpr (kn == 0) ta = tap_addr(buf_off, (tn_ / 3) * p.pitch + (tn_ % 3));
                read_b(ta, kn, bf[(s + 1) & 1]);
            }
            (void)tap; (void)ks;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s % D][j].h, bf[s & 1][i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_w(std::integral_constant<int, s % D>{}, kk0 + s + D);
            if constexpr (s < 7) {
                if (more) issue_halo(std::integral_constant<int, s>{}, csn, nxt_lds);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    }

    // patch-order row -> output row (pixel index over [image][Hv][Wv])
    auto out_row = [&](int m) { return (c.img0 * p.Hv + c.y0 + (m >> p.ltw)) * p.Wv + c.x0 + (m & (p.tw - 1)); };

    if (p.nsplit > 1) {   // fp32 slab of this split for the reduce kernels, straight from registers
        float* base = d.ws + (size_t)c.split * (size_t)d.M * d.N;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = out_row(i * 32 + l31);
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = n_w + j * 32 + 8 * g + 4 * hi;
                    *reinterpret_cast<float4*>(base + (size_t)row * d.N + col) =
                        make_float4(acc[i][j][g * 4], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]);
                }
        }
        return;
    }

    // ---- fused epilogue, part 1 (registers): bias -> act -> * alpha -> fp16 into the LDS tile [128][bn_blk + 8]
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();   // every wave is done with the halo buffers: the tile re-uses that LDS
    asm volatile("" ::: "memory");
    const EpiCtx e = make_epi(d, 0);
    f16* cs = reinterpret_cast<f16*>(smem);
    const int cs_ld = c.bn_blk + 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int lrow = i * 32 + l31;
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int lc = cw + j * 32 + 8 * g + 4 * hi;
                float bq[4] = {0.f, 0.f, 0.f, 0.f};
                if (e.flags & VD_EPI_BIAS) {
                    U2H4 t;
                    t.u = *reinterpret_cast<const uint2*>(e.bias + c.n0_blk + lc);
#pragma unroll
                    for (int q = 0; q < 4; ++q) bq[q] = (float)t.e[q];
                }
                U2H4 o;
                if (e.act == VD_ACT_NONE) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) o.e[q] = (f16)((acc[i][j][g * 4 + q] + bq[q]) * e.alpha);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) o.e[q] = (f16)(apply_act(e.act, acc[i][j][g * 4 + q] + bq[q]) * e.alpha);
                }
                *reinterpret_cast<uint2*>(cs + lrow * cs_ld + lc) = o.u;
            }
    }
}

__global__ __launch_bounds__(256, 1) void conv3x3_wreg_kernel(const WrArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const VdGemmDesc& d = p.g.d;
    WrCtx c;
    c.tid = threadIdx.x;
    c.lane = c.tid & 63;
    c.wave = c.tid >> 6;
    c.hi = c.lane >> 5;
    c.l31 = c.lane & 31;

    // blocks that share a weight panel (same column tile and K split, all patches) get consecutive logical indices inside one
    // XCD's contiguous run; the column tiles of a patch follow each other at distance tiles_m
    const int ntot = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = ntot >> 3, r = ntot & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    c.tm = bid % p.tiles_m;
    const int rest = bid / p.tiles_m;
    const int tn = rest % p.ntn;
    c.split = rest / p.ntn;
    {
        const int tpi = p.tiles_x * p.tiles_y;
        c.img0 = c.tm / tpi;
        const int r = c.tm - c.img0 * tpi;
        const int ty = r / p.tiles_x;
        c.y0 = ty * p.rows;
        c.x0 = (r - ty * p.tiles_x) * p.tw;
    }
    c.c_begin = c.split * p.cps;
    int c_end = c.c_begin + p.cps;
    if (c_end > p.nchunks) c_end = p.nchunks;
    c.ncl = c_end - c.c_begin;   // >= 1 by construction of the launcher
    const int ni0 = p.tn_ni0[tn], ni1 = p.tn_ni1[tn];
    c.n0_blk = p.tn_n0[tn];
    c.bn_blk = (ni0 + ni1) * 64;
    c.rs_a0 = make_rsrc_words(d.a0, p.g.a0_bytes);
    c.rs_a1 = make_rsrc_words(d.a1 ? d.a1 : d.a0, d.a1 ? p.g.a1_bytes : 0u);
    c.lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // halo pieces of this wave: piece q = j * 4 + wave covers halo pixels 8q .. 8q + 7, lane = (pixel in piece) * 8 + physical
    // slot; packed (input pixel << 3 | logical slot), -1 = zeros (padding ring, tail)
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int hp = (j * 4 + c.wave) * 8 + (c.lane >> 3);
        const int hy = (hp * p.mg_pitch) >> 20;
        const int hx = hp - hy * p.pitch;
        const int vy = c.y0 + hy - 1, vx = c.x0 + hx - 1;
        const bool ok = hp < p.hpx && (unsigned)vy < (unsigned)p.Hv && (unsigned)vx < (unsigned)p.Wv;
        const int pix = (c.img0 * d.Hin + (vy >> d.ups)) * d.Win + (vx >> d.ups);
        c.hsrc[j] = ok ? ((pix << 3) | ((c.lane & 7) ^ ((hp >> 1) & 7))) : -1;
    }

    // waves 0-1 own ni0 n tiles each, waves 2-3 ni1 (wave-uniform dispatch to the instance of that width)
    const int wv = __builtin_amdgcn_readfirstlane(c.wave);
    const int my_ni = wv < 2 ? ni0 : ni1;
    const int cw = wv < 2 ? wv * ni0 * 32 : 2 * ni0 * 32 + (wv - 2) * ni1 * 32;
    const int n_w = c.n0_blk + cw;
    switch (my_ni) {
        case 1: wr_wave<1>(p, c, smem, n_w, cw); break;
        case 2: wr_wave<2>(p, c, smem, n_w, cw); break;
        case 3: wr_wave<3>(p, c, smem, n_w, cw); break;
        default: wr_wave<4>(p, c, smem, n_w, cw); break;
    }
    if (p.nsplit > 1) return;

    // ---- part 2: 16-byte row segments of the tile: (+ rowvec) (+ residual) -> global; then the statistics of what was stored
    __syncthreads();
    const EpiCtx e = make_epi(d, 0);
    f16* cs = reinterpret_cast<f16*>(smem);
    const int cs_ld = c.bn_blk + 8;
    const int CH = c.bn_blk / 8;
    const bool want_res = (e.flags & VD_EPI_RESIDUAL) != 0;
    const bool want_rv = (e.flags & VD_EPI_ROWVEC) != 0;
    const bool want_stats = d.out_stats != nullptr && p.g.stat_rows > 0;
    auto out_row = [&](int m) { return (c.img0 * p.Hv + c.y0 + (m >> p.ltw)) * p.Wv + c.x0 + (m & (p.tw - 1)); };
    // the residual / row-vector segments of a batch of 8 segments per thread are requested before any is consumed (the
    // accumulators are dead by now: registers are free, and a serial load -> add -> store chain per segment is what this loop
    // cost in its first version: 47 us of fixed time per 64x64-level launch against 25 for the halo kernel)
    constexpr int EB = 8;
    const int nseg = 128 * CH;
    for (int s0 = c.tid; s0 < nseg; s0 += 256 * EB) {
        uint4 ra[EB], rb[EB];
        int rows_[EB];
#pragma unroll
        for (int k = 0; k < EB; ++k) {
            const int sgm = s0 + k * 256;
            const int sg = sgm < nseg ? sgm : s0;   // the tail re-requests the first segment (discarded)
            const int r = sg / CH, cc = (sg - r * CH) * 8;
            const int col = c.n0_blk + cc;
            const int row = out_row(r);
            rows_[k] = row;
            ra[k] = make_uint4(0, 0, 0, 0);
            rb[k] = make_uint4(0, 0, 0, 0);
            if (want_res) ra[k] = *reinterpret_cast<const uint4*>(e.res + (size_t)row * e.ldr + col);
            if (want_rv) rb[k] = *reinterpret_cast<const uint4*>(e.rowvec + (size_t)(row / e.rows_per_batch) * e.N + col);
        }
#pragma unroll
        for (int k = 0; k < EB; ++k) {
            const int sgm = s0 + k * 256;
            if (sgm < nseg) {
                const int r = sgm / CH, cc = (sgm - r * CH) * 8;
                U4H8 t, a, b, o;
                t.u = *reinterpret_cast<const uint4*>(cs + r * cs_ld + cc);
                a.u = ra[k];
                b.u = rb[k];
#pragma unroll
                for (int q = 0; q < 8; ++q) o.e[q] = (f16)((float)t.e[q] + (float)a.e[q] + (float)b.e[q]);
                f16* dst = reinterpret_cast<f16*>(e.out) + (size_t)rows_[k] * e.ldc + c.n0_blk + cc;
                if (p.g.nt_store) vd_store16_nt(dst, o.u);
                else *reinterpret_cast<uint4*>(dst) = o.u;
                if (want_stats) *reinterpret_cast<uint4*>(cs + r * cs_ld + cc) = o.u;
            }
        }
    }
    if (want_stats) {   // one partial per patch: (mean, M2) of the 128 stored values of every channel of the tile
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem + 128 * cs_ld * 2);   // [8 row lanes][bn_blk][2]
        const int OCT = CH;                     // channel octets of the tile (<= 64)
        const int LANES = 256 / OCT < 8 ? 256 / OCT : 8;
        const int co = c.tid % OCT, rl = c.tid / OCT;
        if (rl < LANES) {
            U4H8 kk;
            kk.u = *reinterpret_cast<const uint4*>(cs + co * 8);
            float S[8], Q[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) S[q] = Q[q] = 0.f;
            for (int r = rl; r < 128; r += LANES) {
                U4H8 t;
                t.u = *reinterpret_cast<const uint4*>(cs + r * cs_ld + co * 8);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float v = (float)t.e[q] - (float)kk.e[q];
                    S[q] += v;
                    Q[q] = fmaf(v, v, Q[q]);
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) *reinterpret_cast<float2*>(red + (rl * c.bn_blk + co * 8 + q) * 2) = make_float2(S[q], Q[q]);
        }
        __syncthreads();
        for (int ch = c.tid; ch < c.bn_blk; ch += 256) {
            float S = 0.f, Q = 0.f;
            for (int l = 0; l < LANES; ++l) {
                const float2 v = *reinterpret_cast<const float2*>(red + (l * c.bn_blk + ch) * 2);
                S += v.x;
                Q += v.y;
            }
            *reinterpret_cast<float2*>(d.out_stats + ((size_t)c.tm * d.N + c.n0_blk + ch) * 2) =
                make_float2((float)cs[ch] + S / 128.f, fmaxf(Q - S * S / 128.f, 0.f));
        }
    }
}

}  // namespace
