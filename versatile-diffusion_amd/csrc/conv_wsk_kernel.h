// 3x3 convolution of the 8x8 level WITHOUT a split over blocks (round 5): conv3x3_wstream_kernel (conv_wstream_kernel.h) splits
// K ten ways over blocks, leaves 26 MB of fp32 slabs and needs a reduce launch for the epilogue -- 28 + 11 us per layer of which
// ~10 us are MFMAs.  Here a block owns 128 pixels (2 images) x 32 output channels for the WHOLE K and its four waves split the
// k-steps of every 64-channel chunk among them (wave w takes the channels 16 w .. 16 w + 15 of the chunk at all nine taps):
//   * 4 row tiles x N / 32 column tiles = 160 blocks at the bench shape, one per CU; 720 MFMAs per wave;
//   * weights as in conv3x3_wstream_kernel: MFMA-fragment order ([n tile][chunk][tap][k-step][lane][8], the same packed tensor),
//     one coalesced 1-KiB global load per wave and k-step straight into registers, a ring of 36 fragments (four chunks) in flight;
//   * the 2 x (10 x 10) pixel halo of a chunk by LDS-DMA into a ring of FOUR buffers, requested three chunks (~1.6 us) ahead,
//     one barrier per chunk; all four waves read the same halo (different 16-channel slices of it).  (LDS-DMA goes through
//     inline asm, so hipcc's vmcnt for a weight fragment also waits for the halo pieces issued behind it: the ring of 36
//     fragments is effectively ~20 deep.  A fifth, halo-only wave restores the 36 and measured SLOWER at 160 blocks -- 39-42 us
//     against 28 us -- and equal at 80: the launch is bound by the ~27 GB/s a CU ingests through its L1, not by latency);
//   * the chunk sequence is unrolled behind forward guards (gemm_wstream_kernel.h explains why: with a back edge hipcc drains
//     the fragment ring once per trip);
//   * the partial sums of the four waves meet in LDS (fp32 [4][128][36]); the fused epilogue (bias, row vector, residual,
//     activation, per-channel statistics of the stored tile for the consuming GroupNorm) runs in the kernel: no slabs, no
//     reduce launch, no inter-block synchronisation.
// Replaces the same reference lines as conv3x3_wstream_kernel (/root/reference/lib/model_zoo/openaimodel.py:254-274 at ds = 8).
// Included by conv_wstream.hip; the folded skip convolution stays on the split kernel.
#pragma once
#include "conv_wstream_kernel.h"

namespace {

struct WkArgs {
    WsArgs w;
    GemmArgs g;   // normalised descriptor: epilogue operands, out_stats
    int rotate;   // the image groups of a column tile start at different points of the chunk ring
};

constexpr int WK_NB = 4;                      // halo buffers
constexpr int WK_PF = WK_NB - 1;             // chunks a halo is requested ahead
constexpr int WK_D = 36;                     // weight fragments in flight per wave = 4 chunks x 9 k-steps
constexpr int WK_MAXC = 40;                  // unrolled chunks (2560 input channels)
constexpr int WK_RP = 36;                    // fp32 pitch of the partial-sum tiles
constexpr int WK_LDS_MAIN = WK_NB * WS_HB;
constexpr int WK_LDS_RED = 4 * 128 * WK_RP * 4;
constexpr int WK_CS_LD = 40;                 // fp16 pitch of the stored tile (statistics pass)
constexpr int WK_LDS = (WK_LDS_MAIN > WK_LDS_RED + 128 * WK_CS_LD * 2) ? WK_LDS_MAIN : (WK_LDS_RED + 128 * WK_CS_LD * 2);
static_assert(WK_LDS <= 160 * 1024, "LDS budget");

constexpr int WK_NT = 256;

__global__ __launch_bounds__(WK_NT, 1) void conv3x3_wsk_kernel(const WkArgs q) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const WsArgs& p = q.w;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int hi = lane >> 5, l31 = lane & 31;

    // the tiles_m image groups of one column tile get consecutive logical indices inside one XCD's run: its weight panel comes
    // from HBM once and hits that L2 for the other groups
    const int ntot = gridDim.x;
    int bid = blockIdx.x;
    {
        const int qq = ntot >> 3, r = ntot & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + idx;
    }
    const int tm = bid % p.tiles_m;
    const int tn = bid / p.tiles_m;
    const int img0 = tm * WS_IPB;
    const int n0 = tn * 32;
    const int ncl = p.nchunks;
    // The tiles_m blocks of a column tile stream the SAME weight panel.  In lock-step they would all miss on the same lines and
    // the panel would arrive at ONE CU's outstanding-miss rate (~25 GB/s: measured 28 us per launch whatever the grid size);
    // each block therefore walks the chunk ring from its own starting point (rot): at any time the blocks of a panel fetch
    // different chunks from HBM and find the others' in the XCD's L2.  (Changes the fp32 accumulation order per image group,
    // not the determinism.)
    const int rot = q.rotate ? (tm * ncl) / p.tiles_m : 0;
    auto chunk_at = [&](int sp) {   // sequence position (clamped: the tail re-reads the last one) -> chunk
        int c = (sp < ncl ? sp : ncl - 1) + rot;
        return c >= ncl ? c - ncl : c;
    };

    const i32x4 rs_a0 = make_rsrc_words(p.a0, p.a0_bytes);
    const i32x4 rs_a1 = make_rsrc_words(p.a1 ? p.a1 : p.a0, p.a1 ? p.a1_bytes : 0u);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // ---- halo pieces of this wave (as conv3x3_wstream_kernel): piece j * 4 + wave covers 8 halo pixels, lane = pixel * 8 + slot
    int hsrc[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int hp = (j * 4 + wave) * 8 + (lane >> 3);
        const int g = hp >= WS_GPX ? 1 : 0;
        const int rem = hp - g * WS_GPX;
        const int yh = (rem * 373) >> 12;        // rem / 11 for rem < 128
        const int xh = rem - yh * WS_PITCH;
        const bool ok = hp < WS_IPB * WS_GPX && yh >= 1 && yh <= 8 && xh >= 1 && xh <= 8;
        const int pix = ((img0 + g) * 8 + yh - 1) * 8 + xh - 1;
        hsrc[j] = ok ? ((pix << 3) | ((lane & 7) ^ (xh & 7))) : -1;
    }
    struct ChunkSrc { i32x4 rs; int ld2; unsigned soff; };
    auto chunk_src = [&](int sp) {   // sp: position in this block's chunk sequence
        ChunkSrc s;
        const int cc = chunk_at(sp) * 64;
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

    // ---- weight stream of this wave: n tile n0 / 32, k-step (tap, ks = wave) of every chunk: fragment (chunk * 36 + tap * 4 + wave)
    const uint4* wq = p.wp + ((size_t)(n0 >> 5) * ncl * 36 + wave_s) * 64 + lane;
    U4H8 wf[WK_D];
    auto load_w = [&](auto st, int sp, auto tt) {   // slot, sequence position, tap
        constexpr int slot = decltype(st)::value, tap = decltype(tt)::value;
        wf[slot].u = wq[((size_t)chunk_at(sp) * 36 + tap * 4) * 64];
    };

    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // pixel fragments (conv3x3_wstream_kernel): byte = hp0b[i] + (tkx[kx] ^ (ks << 5)) + (ky * 11 + kx) * 128, ks = wave
    int hp0b[4], tkxw[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = i * 32 + l31;
        hp0b[i] = ((m >> 6) * WS_GPX + ((m >> 3) & 7) * WS_PITCH + (m & 7)) * 128;
    }
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) tkxw[kx] = (((((l31 & 7) + kx) & 7) ^ hi) << 4) ^ ((wave & 3) << 5);
    auto read_b = [&](auto tt, int buf_off, f16x8* b) {
        constexpr int tap = decltype(tt)::value, ky = tap / 3, kx = tap % 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            U4H8 v;
            v.u = *reinterpret_cast<const uint4*>(smem + buf_off + hp0b[i] + tkxw[kx] + (ky * WS_PITCH + kx) * 128);
            b[i] = v.h;
        }
    };

    // ---- prologue: halos of the first three chunks, the first 36 weight fragments (chunks 0 .. 3)
    ws_static_for<0, WK_PF>([&](auto ct) {
        constexpr int c = decltype(ct)::value;
        const ChunkSrc cs = chunk_src(c);
        ws_static_for<0, 7>([&](auto jt) { issue_halo(jt, cs, lds0 + (unsigned)(c * WS_HB)); });
    });
    ws_static_for<0, 4>([&](auto ct) {
        ws_static_for<0, 9>([&](auto tt) {
            load_w(std::integral_constant<int, decltype(ct)::value * 9 + decltype(tt)::value>{}, decltype(ct)::value, tt);
        });
    });

    f16x8 bf[2][4];
    auto chunk = [&](auto lt) {
        constexpr int lc = decltype(lt)::value;
        constexpr int buf_off = (lc % WK_NB) * WS_HB;
        const unsigned nxt_lds = lds0 + (unsigned)(((lc + WK_PF) % WK_NB) * WS_HB);
        const ChunkSrc csn = chunk_src(lc + WK_PF);
        // halo(lc) has landed for this wave: the requests issued behind its last piece are the 14 pieces of two more halos and
        // the weight loads in between (prologue: 7 pieces x (2 - lc) + 36 loads + 16 per chunk done; steady state: 2 + 2 x 16)
        constexpr int younger = lc >= WK_PF ? 2 + 16 * (WK_PF - 1) : (7 * (WK_PF - 1 - lc) + WK_D + 16 * lc);
        wait_vm<(younger > 63 ? 63 : younger)>();
        __builtin_amdgcn_s_barrier();   // ... for every wave; every wave has left chunk lc - 1, whose buffer the DMA below refills
        asm volatile("" ::: "memory");
        read_b(std::integral_constant<int, 0>{}, buf_off, bf[0]);
        ws_static_for<0, 9>([&](auto tt) {
            constexpr int t = decltype(tt)::value;
            if constexpr (t < 8) read_b(std::integral_constant<int, t + 1>{}, buf_off, bf[(t + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[(lc % 4) * 9 + t].h, bf[t & 1][i], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            load_w(std::integral_constant<int, (lc % 4) * 9 + t>{}, lc + 4, tt);   // always issued: every step adds exactly one load
            if constexpr (t < 7) issue_halo(tt, csn, nxt_lds);                      // ... and (t < 7) one halo piece
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    ws_static_for<0, WK_MAXC>([&](auto lt) {
        if (decltype(lt)::value < ncl) chunk(lt);
    });

    // ---- the four partial sums meet in LDS; fused epilogue
    const VdGemmDesc& d = q.g.d;
    const EpiCtx e = make_epi(d, 0);
    const bool worker = true;
    const int er = (tid & 255) >> 1, ec = (tid & 1) * 16;  // this thread's 16 output columns of tile row er
    const int grow = img0 * 64 + er, gcol = n0 + ec;
    // epilogue operands requested before the drain: bias, row vector, residual (two 16-byte pieces each; absent = zeros)
    uint4 ob[2], ov[2], orr[2];
    {
        const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(e.bias), 0, (e.flags & VD_EPI_BIAS) ? e.N * 2 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(e.rowvec), 0, (e.flags & VD_EPI_ROWVEC) ? 0x7fffffff : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16*>(e.res), 0, (e.flags & VD_EPI_RESIDUAL) ? 0x7fffffff : 0, 0x00020000);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const vd_u32x4_t b = __builtin_amdgcn_raw_buffer_load_b128(rs_b, (gcol + 8 * h) * 2, 0, 0);
            const vd_u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs_v, ((grow / e.rows_per_batch) * e.N + gcol + 8 * h) * 2, 0, 0);
            const vd_u32x4_t r = __builtin_amdgcn_raw_buffer_load_b128(rs_r, (grow * e.ldr + gcol + 8 * h) * 2, 0, 0);
            ob[h] = make_uint4(b[0], b[1], b[2], b[3]);
            ov[h] = make_uint4(v[0], v[1], v[2], v[3]);
            orr[h] = make_uint4(r[0], r[1], r[2], r[3]);
        }
    }
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();   // every wave is done with the halo buffers: the partial sums re-use that LDS
    asm volatile("" ::: "memory");
    float* red = reinterpret_cast<float*>(smem);
    if (worker) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(red + ((size_t)(wave * 128 + i * 32 + l31) * WK_RP + 8 * g + 4 * hi)) =
                    make_float4(acc[i][g * 4], acc[i][g * 4 + 1], acc[i][g * 4 + 2], acc[i][g * 4 + 3]);
    }
    __syncthreads();
    f16* cs = reinterpret_cast<f16*>(smem + WK_LDS_RED);
    if (worker) {
        float v[16];
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            float4 s = *reinterpret_cast<const float4*>(red + (size_t)er * WK_RP + ec + 4 * c4);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float4 t = *reinterpret_cast<const float4*>(red + (size_t)(w * 128 + er) * WK_RP + ec + 4 * c4);
                s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
            }
            v[4 * c4] = s.x; v[4 * c4 + 1] = s.y; v[4 * c4 + 2] = s.z; v[4 * c4 + 3] = s.w;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            U4H8 b, rv, rs, o;
            b.u = ob[h]; rv.u = ov[h]; rs.u = orr[h];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float x = apply_act(e.act, v[8 * h + i] + (float)b.e[i]) * e.alpha;
                x += (float)rv.e[i];
                x += (float)rs.e[i];
                o.e[i] = (f16)x;
            }
            *reinterpret_cast<uint4*>(reinterpret_cast<f16*>(e.out) + (size_t)grow * e.ldc + gcol + 8 * h) = o.u;
            *reinterpret_cast<uint4*>(cs + er * WK_CS_LD + ec + 8 * h) = o.u;
        }
    }
    // ---- per-channel statistics of the stored tile, one partial per image (64 rows): out_stats[(image) * N + channel]
    if (d.out_stats != nullptr) {
        __syncthreads();
        if (tid < 64) {
            const int img = tid >> 5, ch = tid & 31;
            const f16* col = cs + (img * 64) * WK_CS_LD + ch;
            const float k0 = (float)col[0];
            float S = 0.f, Q = 0.f;
#pragma unroll 8
            for (int r = 0; r < 64; ++r) {
                const float x = (float)col[r * WK_CS_LD] - k0;
                S += x;
                Q = fmaf(x, x, Q);
            }
            *reinterpret_cast<float2*>(d.out_stats + ((size_t)(img0 + img) * d.N + n0 + ch) * 2) =
                make_float2(k0 + S * (1.0f / 64.0f), fmaxf(Q - S * S * (1.0f / 64.0f), 0.f));
        }
    }
}

}  // namespace
