// Weight-streaming GEMM for the long-K, small-M projections of the UNet's deep levels (round 4): y = x W^T with M = 2048 / 512
// rows against N x K = 1280 x 5120 -- the output projection of the gated feed-forward at the 16x16 / 8x8 levels
// (/root/reference/lib/model_zoo/attention.py:37-64: FeedForward.net[2]).  Included by conv_wstream.hip (vd_gemm_wstream_f16).
//
// gemm_f16_kernel brings BOTH operands of a 128 x 128 tile through LDS by LDS-DMA and spends 0.77 us per 64-deep k-step with
// 0.43 us of MFMAs in it (tools/probes/gemm_timeline.py).  The weights of a layer are static: here they are packed once in
// MFMA-fragment order (vd_hip/pack.py: pack_linear_weight_stream, [N / 32][K / 64][4 k-steps][64 lanes][8 halfs]) and go
// global -> registers as fully coalesced 1-KiB loads, two chunks (8 k-steps) ahead; only the activation tile
// ([128 rows][64 k] = 16 KiB per chunk, double-buffered, one barrier per chunk) takes the LDS path.  conv3x3_wstream_kernel
// without the taps: a block = 4 waves x 64 output columns over the same 128 rows, 8 MFMAs per k-step and wave from 2 weight
// fragments (registers) + 4 row fragments (LDS).  K is split over 64-deep chunks across blocks; the fp32 slabs go to the split-K
// reduce kernels of gemm.hip, which run the fused epilogue (bias, residual, statistics).
#pragma once
#include "gemm_kernel.h"

namespace {

struct GwArgs {
    const f16* a; const uint4* wp; float* ws;
    int lda, M, N;
    int nchunks, cps;          // 64-deep chunks of K: in total / per split
    int tiles_m, tiles_n, nsplit;
    unsigned a_bytes;
};

constexpr int GW_TILE = 128 * 128;   // bytes of one activation tile [128 rows][64 halfs]
constexpr int GW_MAXC = 32;          // chunks per block (the chunk sequence is unrolled)

template <int LO, int HI, class F>
__device__ __forceinline__ void gw_static_for(F&& f) {
    if constexpr (LO < HI) {
        f(std::integral_constant<int, LO>{});
        gw_static_for<LO + 1, HI>(f);
    }
}

__global__ __launch_bounds__(256, 1) void gemm_wstream_kernel(const GwArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int hi = lane >> 5, l31 = lane & 31;

    // blocks that share a weight panel (same column slice and K split, the tiles_m row tiles) get consecutive logical indices
    // inside one XCD's contiguous run: the panel comes from HBM once and hits that L2 for the other row tiles
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
    const int m0 = tm * 128;
    const int n0 = tn * 256 + wave_s * 64;
    const int c_begin = split * p.cps;
    int c_end = c_begin + p.cps;
    if (c_end > p.nchunks) c_end = p.nchunks;
    const int ncl = c_end - c_begin;   // >= 1 by construction of the launcher

    const i32x4 rs_a = make_rsrc_words(p.a, p.a_bytes);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // ---- activation tile of a chunk: 16 pieces of 8 rows x 128 bytes, wave w issues pieces w, w + 4, w + 8, w + 12.  The DMA
    // destination is lane-linear (row = 8 q + lane / 8, physical slot = lane % 8), so the XOR swizzle of the LDS image
    // (lds_off_kb<64>) is applied on the source side: the lane fetches the logical slot that lives at its physical slot
    unsigned avoff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (j * 4 + wave) * 8 + (lane >> 3);
        const int row = m0 + r;
        avoff[j] = row < p.M ? (unsigned)((row * p.lda + (((lane & 7) ^ lds_swz<64>(r)) << 3)) * 2) : OOB_OFFSET;
    }
    auto issue_a = [&](int c, int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            dma16(rs_a, lds0 + (unsigned)(buf * GW_TILE + (j * 4 + wave_s) * 1024), avoff[j], (unsigned)(c * 128));
    };

    // ---- weight stream of this wave: two n tiles, 4 fragments (1 KiB each) per chunk and tile, chunks contiguous
    const int nt0 = n0 >> 5;
    const uint4* wq0 = p.wp + (size_t)nt0 * p.nchunks * 256 + lane;
    const uint4* wq1 = p.wp + (size_t)(nt0 + 1) * p.nchunks * 256 + lane;
    const int c_last = c_end - 1;
    U4H8 wf[2][4][2];   // [chunk parity][k-step][n tile]
    auto load_w = [&](auto pt, auto st, int c) {   // fragments of k-step s of chunk c (clamped: the tail re-reads the last chunk)
        constexpr int par = decltype(pt)::value, s = decltype(st)::value;
        const int cc = c < c_last ? c : c_last;
        wf[par][s][0].u = wq0[((size_t)cc * 4 + s) * 64];
        wf[par][s][1].u = wq1[((size_t)cc * 4 + s) * 64];
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // row fragments: lane row m = i * 32 + l31 of the block's 128, logical slot 2 ks + hi (the swizzle key of row i * 32 + l31
    // is that of l31)
    int rd[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) rd[ks] = lds_off_kb<64>(l31, ks * 2 + hi);

    // ---- prologue: activation tile of the first chunk, weights of the first two chunks
    issue_a(c_begin, 0);
    gw_static_for<0, 4>([&](auto st) { load_w(std::integral_constant<int, 0>{}, st, c_begin); });
    gw_static_for<0, 4>([&](auto st) { load_w(std::integral_constant<int, 1>{}, st, c_begin + 1); });

    auto chunk = [&](auto pt, int lc) {
        constexpr int par = decltype(pt)::value;
        const int c = c_begin + lc;
        // this chunk's activation tile (requested during the previous chunk, in front of the 8 weight loads of chunk c + 1) and
        // its weights (requested two chunks ago) have landed for this wave ... for every wave; every wave has left chunk c - 1,
        // whose buffer the DMA below refills
        wait_vm<8>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (lc + 1 < ncl) issue_a(c + 1, par ^ 1);
        const char* st = smem + par * GW_TILE;
        gw_static_for<0, 4>([&](auto kt) {
            constexpr int ks = decltype(kt)::value;
            f16x8 bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                U4H8 v;
                v.u = *reinterpret_cast<const uint4*>(st + rd[ks] + i * 32 * 128);
                bf[i] = v.h;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[par][ks][j].h, bf[i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_w(pt, kt, c + 2);   // always issued (clamped): every chunk adds exactly 8 weight loads behind its DMA pieces
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    // No loop: the chunks of a block (at most GW_MAXC, the launcher splits K accordingly) are unrolled behind forward guards.
    // With a back edge hipcc cannot count the weight loads that are in flight ACROSS iterations and puts s_waitcnt vmcnt(0) in
    // front of the first MFMA of every iteration -- the ring drained once per trip (conv3x3_wstream_kernel pays that once per
    // 36-k-step chunk; here it would be every second 4-k-step chunk).
    gw_static_for<0, GW_MAXC>([&](auto lt) {
        constexpr int lc = decltype(lt)::value;
        if (lc < ncl) chunk(std::integral_constant<int, lc & 1>{}, lc);
    });

    // ---- fp32 slab of this split for the reduce kernel, straight from registers (4 consecutive floats per lane and group)
    float* base = p.ws + (size_t)split * (size_t)p.M * p.N;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = m0 + i * 32 + l31;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = n0 + j * 32 + 8 * g + 4 * hi;
                if (row < p.M)
                    *reinterpret_cast<float4*>(base + (size_t)row * p.N + col) =
                        make_float4(acc[i][j][g * 4], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]);
            }
    }
}

}  // namespace
