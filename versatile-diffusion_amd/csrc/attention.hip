// Fused attention forward for gfx950: softmax(Q K^T * scale) V with online softmax.
// The (B*heads, Nq, Nk) score tensor the reference materialises three times over
// (sim*scale -> softmax -> einsum, lib/model_zoo/attention.py:176-191; 2.1 GB per self-attention
// layer at 64x64 latents, bs 4 + CFG) never leaves registers here.
//
// Mapping (64-wide wavefronts, v_mfma_f32_32x32x16_f16):
//   block = 4 waves, wave w owns 32 query rows; K/V tiles of 64 keys are staged in LDS for the block.
//   S^T = K Q^T is computed "swapped" (A = K rows from LDS, B = Q rows held in registers), so a lane
//   owns ONE query (column lane&31) and 16 keys per 32-key tile: the softmax row reduction is 15
//   in-lane max/add plus a single cross-half exchange (lane ^ 32), no LDS traffic.
//   The C/D register layout of that MFMA (row = (r&3) + 8*(r>>2) + 4*(lane>>5)) is exactly the
//   k-grouping a B operand wants (8 k-values per lane-half), so P goes from accumulator to the
//   B operand of O^T += V^T P^T as a plain fp16 convert - no shuffles, no LDS round trip.  V is
//   transposed while it is staged (LDS holds V^T[d][key]) so the A operand is two ds_read_b64.
//   O^T accumulators: lane owns its query column again, so rescaling by exp(m_old - m_new) is
//   lane-local as well.
#include <atomic>
#include <stdlib.h>
#include <type_traits>
#include "vd_common.h"
#include "../../include/vd_hip.h"

namespace {

#ifndef VD_ATTN_MINW
#define VD_ATTN_MINW 1
#endif
#ifndef VD_ATTN_W8_MINW
#define VD_ATTN_W8_MINW 1
#endif
constexpr int KV = 64;    // keys per tile
constexpr float RESCALE_THR = 6.0f;  // log2 units: P values stay <= 64 between rescales

struct AttnArgs {
    const f16* q;
    const f16* k;
    const f16* v;
    f16* o;
    int H, Nq, Nk, ldq, ldk, ldv, ldo;
    int64_t sq, sk, sv, so;
    float scale_log2;
    int causal;
    int nqb, BH;
    int ctx_map;   // block -> (batch, head, query block) mapping for short contexts, see the kernel
    int pair_contig;   // long K / V: an XCD owns consecutive (batch, head) pairs (else pairs xcd, xcd + 8, ...)
};

// NWV waves per block (4, or 8 for long self-attention: the K / V tile's LDS-DMA requests are shared by twice the queries, and
// request issue is serial time in a wave -- 4 pieces per tile and wave become 2).
template <int D, int NWV = 4>
__global__ __launch_bounds__(64 * NWV, (NWV == 8 ? VD_ATTN_W8_MINW : (D <= 64 && NWV == 4 ? VD_ATTN_MINW : 1))) void attn_fwd_kernel(const AttnArgs p) {
    constexpr int QB = 32 * NWV;            // queries per block
    constexpr int KS = (D + 15) / 16;       // k-steps of the QK^T MFMA
    constexpr int DB = (D + 31) / 32;       // 32-row blocks of O^T == 32-column panels of the V image
    constexpr int CPR = 2 * KS + 1;         // 16-byte chunks per K row in LDS: data, zero pad to KS*16, +1 (odd stride)
    constexpr int KROW = CPR * 8;           // K row stride in halfs
    constexpr int K_BYTES = KV * CPR * 16;  // K image: [64 keys][CPR chunks]
    constexpr int V_BYTES = DB * KV * 64;   // V image: DB panels of [64 keys][32 halfs] (64-byte rows)
    constexpr int TILE_BYTES = K_BYTES + V_BYTES;
    static_assert(D % 8 == 0, "head dim must be a multiple of 8");
    // double-buffered tiles (one barrier per tile) where two copies fit the 64 KiB static LDS limit
    constexpr int NBUF = (2 * TILE_BYTES <= 60 * 1024) ? 2 : 1;
    constexpr bool HAS_ONES = (D % 32) != 0;  // spare rows in the last O^T block: row sums ride on the P.V MFMA (see below)
    constexpr int ONES_COL = D % 32;          // column inside panel DB-1 (a multiple of 8: first half of a 16-byte chunk)

    __shared__ __attribute__((aligned(1024))) char lds_all[NBUF * TILE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;

    // XCD-aware mapping: keep all query blocks of one (batch, head) on one XCD so K/V stay in its L2
    int qb, bh;
    {
        const int bid = blockIdx.x;
        if (p.ctx_map) {
            // short context (cross-attention, Nk of one or two tiles): K / V re-use across query blocks is worthless, what
            // costs is that the H heads of a query row each touch D * 2 bytes of the same cache lines of q and out.  Run the
            // heads of one (batch, query block) back to back on ONE XCD: the partial-line stores merge in its L2 and q
            // lines are fetched once.  (The number of (batch, query block) groups must be a multiple of 8.)
            const int xcd = bid & 7, idx = bid >> 3;
            const int grp = xcd + 8 * (idx / p.H);
            bh = (grp / p.nqb) * p.H + idx % p.H;
            qb = grp % p.nqb;
        } else if ((p.BH & 7) == 0) {
            // an XCD owns BH / 8 CONSECUTIVE (batch, head) pairs (round 5; before: pairs xcd, xcd + 8, ... = one head of every
            // sample): the heads of a token are neighbours in memory (D * 2 = 80 .. 320 bytes each inside the fused q|k|v row),
            // so consecutive pairs share the 128-byte lines of K / V / q / out inside ONE L2 instead of pulling every line into
            // two of them (counter traffic of the D = 40 launch: 2.1x the unique bytes).
            const int xcd = bid & 7, idx = bid >> 3;
            bh = p.pair_contig ? xcd * (p.BH >> 3) + idx / p.nqb : xcd + 8 * (idx / p.nqb);
            qb = idx % p.nqb;
        } else {
            bh = bid / p.nqb;
            qb = bid % p.nqb;
        }
    }
    const int b = bh / p.H, h = bh % p.H;
    const f16* qp = p.q + (size_t)b * p.sq + h * D;
    const f16* kp = p.k + (size_t)b * p.sk + h * D;
    const f16* vp = p.v + (size_t)b * p.sv + h * D;
    f16* op = p.o + (size_t)b * p.so + h * D;

    // ---- staging plan.  K and V tiles go global -> LDS by DMA (buffer_load ... lds): no VGPR round trip, no ds_write,
    // no transpose pass.  A DMA instruction writes lane * 16 bytes linearly, so the LDS image is chosen by WHICH 16 bytes
    // each lane fetches:
    //   K image  [key][CPR chunks]         chunk q -> key q / CPR, slot q % CPR (slots past D fetch out of range = zeros)
    //   V image  [panel][key][32 halfs]    chunk q -> panel q >> 8, key (q >> 2) & 63, 8 channels (q & 3) of the panel
    // Rows past Nk fall outside the buffer descriptor and read as zeros (the tile offset lives in the VGPR offset: the
    // SGPR offset is not range checked).  V stays row-major; ds_read_b64_tr_b16 transposes it on the way to the MFMA.
    const i32x4 rs_k = make_rsrc_words(kp, (unsigned)(((size_t)(p.Nk - 1) * p.ldk + D) * 2));
    const i32x4 rs_v = make_rsrc_words(vp, (unsigned)(((size_t)(p.Nk - 1) * p.ldv + D) * 2));
    constexpr unsigned OOB = 0x80000000u;
    // wave w issues K instructions w, w+4, ... (< CPR) and V instructions w, w+4, ... (4*DB of them: DB per wave)
    constexpr int KM = (CPR + NWV - 1) / NWV;
    constexpr int VM = (4 * DB + NWV - 1) / NWV;   // V pieces (4 per panel) per wave
    unsigned voff_k[KM], voff_v[VM];
#pragma unroll
    for (int m = 0; m < KM; ++m) {
        const int q = (wave + NWV * m) * 64 + lane;
        const int r = q / CPR, slot = q - r * CPR;
        voff_k[m] = (slot * 8 < D) ? (unsigned)((r * p.ldk + slot * 8) * 2) : OOB;
    }
#pragma unroll
    for (int m = 0; m < VM; ++m) {
        const int q = (wave + NWV * m) * 64 + lane;
        const int d0 = (q >> 8) * 32 + (q & 3) * 8;
        voff_v[m] = (d0 < D && (wave + NWV * m) < 4 * DB) ? (unsigned)((((q >> 2) & 63) * p.ldv + d0) * 2) : OOB;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_all;
    const unsigned k_tile_stride = (unsigned)(KV * p.ldk * 2), v_tile_stride = (unsigned)(KV * p.ldv * 2);
    auto stage = [&](int t, int buf) {
        const unsigned dst = lds0 + (unsigned)(buf * TILE_BYTES) + (unsigned)(wave * 1024);
        const unsigned kt_off = (unsigned)t * k_tile_stride, vt_off = (unsigned)t * v_tile_stride;
#pragma unroll
        for (int m = 0; m < KM; ++m)
            if (m < CPR / NWV || wave < CPR % NWV) dma16(rs_k, dst + m * (NWV * 1024), voff_k[m] + kt_off, 0);
#pragma unroll
        for (int m = 0; m < VM; ++m)
            if (m < (4 * DB) / NWV || wave < (4 * DB) % NWV) dma16(rs_v, dst + K_BYTES + m * (NWV * 1024), voff_v[m] + vt_off, 0);
    };

    // lanes whose V chunk starts at column D of the last panel plant the 1.0 after their own DMA has landed
    bool ones_lane[VM];
#pragma unroll
    for (int m = 0; m < VM; ++m) {
        const int q = (wave + NWV * m) * 64 + lane;
        ones_lane[m] = HAS_ONES && (wave + NWV * m) < 4 * DB && (q >> 8) == DB - 1 && (q & 3) == ONES_COL / 8;
    }
    auto plant_ones = [&](int buf) {
        if constexpr (HAS_ONES) {
#pragma unroll
            for (int m = 0; m < VM; ++m)
                if (ones_lane[m])
                    *reinterpret_cast<f16*>(lds_all + buf * TILE_BYTES + K_BYTES + (wave + NWV * m) * 1024 + lane * 16) = (f16)1.0f;
        }
    };

    // ---- Q fragments: B operand, lane = (query l31, k-half hi), 8 consecutive d per k-step
    const int qrow = qb * QB + wave * 32 + l31;
    f16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int d0 = ks * 16 + hi * 8;
        U4H8 t;
        t.u = make_uint4(0, 0, 0, 0);
        if (qrow < p.Nq && d0 < D) t.u = *reinterpret_cast<const uint4*>(qp + (size_t)qrow * p.ldq + d0);
        // softmax scale and log2(e) are folded into Q once: the scores leave the MFMA already in exp2 units
#pragma unroll
        for (int j = 0; j < 8; ++j) t.e[j] = (f16)((float)t.e[j] * p.scale_log2);
        qf[ks] = t.h;
    }

    f32x16 acc[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    // Running max m of each query (log2 units) lives as -m in all 16 slots of `negm`, the C operand of the first QK^T MFMA:
    // the matrix core subtracts it for free and S' = s - m comes out of the MFMA ready for exp2.  m lags behind the
    // true row max by at most RESCALE_THR (deferred rescale), so P = exp2(S') <= 2^RESCALE_THR between rescales.
    f32x16 negm;
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[r] = 0.f;
    // Row sums: when the head dim leaves spare rows in the last 32-row block of O^T (D = 40, 80), V gets a column of
    // ones at d = D, so the P.V MFMA accumulates sum_k P[k] in accumulator row D -- no VALU adds, and the deferred
    // rescale covers it with the rest of the accumulator.
    float l_run = 0.f;                // VALU row sum, only when !HAS_ONES

    int ntiles = (p.Nk + KV - 1) / KV;
    if (p.causal) {
        const int last_q = min(qb * QB + QB - 1, p.Nq - 1);
        ntiles = min(ntiles, (last_q + p.causal - 1) / KV + 1);  // causal = 1 + offset: key <= query + offset is visible
    }

    // per-lane LDS byte offsets of the operand reads inside a tile
    const unsigned k_lane = (unsigned)((l31 * KROW + hi * 8) * 2);
    // transpose read: in each 16-lane group lane i addresses row (i >> 2), column quad (i & 3) of a [4 keys][16 d] block
    // and receives column i; groups = (d half of the panel, key half hi)
    const unsigned v_lane = (unsigned)(K_BYTES + (4 * hi + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8);
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4* lds_h4_ptr;

    {
        stage(0, 0);
        // The Q loads are the only VMEM results the compiler tracks: consume them here so its s_waitcnt vmcnt(0) lands before
        // the loop.  (It cannot see the hand-written waits; left pending, it would drain vmcnt -- and with it the DMA just
        // issued for the next tile -- in front of the first MFMA of every iteration.)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" ::"v"(qf[ks]));
        wait_vmcnt<0>();
        plant_ones(0);
        __syncthreads();
    }

    // ---- the three phases of one K/V tile (slot = ring slot of the tile in LDS)
    f32x16 st[KV / 32];
    f16x8 pb[KV / 32][2];
    // S^T tiles (keys x queries): S' = s - m (the running max rides in as the C operand)
    auto qk = [&](const int slot) __attribute__((always_inline)) {
        const f16* Ks = reinterpret_cast<const f16*>(lds_all + slot * TILE_BYTES);
#pragma unroll
        for (int kt = 0; kt < KV / 32; ++kt) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                U4H8 a;
                a.u = *reinterpret_cast<const uint4*>(Ks + (kt * 32 + l31) * KROW + ks * 16 + hi * 8);
                st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, qf[ks], ks == 0 ? negm : st[kt], 0, 0, 0);
            }
        }
    };
    // online softmax for query `qrow`; this lane sees keys key0 + kt*32 + (r&3)+8*(r>>2)+4*hi
    auto softmax = [&](const int t) __attribute__((always_inline)) {
        const int key0 = t * KV;
        // masking is needed only on the ragged last tile / on tiles that cross the causal diagonal (wave-uniform)
        const bool need_mask = (key0 + KV > p.Nk) || (p.causal && (key0 + KV - 1 > qb * QB + wave * 32 + p.causal - 1));
        if (need_mask) {
#pragma unroll
            for (int kt = 0; kt < KV / 32; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key >= p.Nk || (p.causal && key > qrow + p.causal - 1)) st[kt][r] = -INFINITY;
                }
        }
        float mx = st[0][0];
#pragma unroll
        for (int kt = 0; kt < KV / 32; ++kt)
#pragma unroll
            for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, st[kt][r]), st[kt][r + 1]);  // v_max3_f32
        // Deferred rescale: accumulators and running max move only when some row grew by more than RESCALE_THR (then every
        // lane updates exactly); otherwise P = exp2(S') <= 2^RESCALE_THR, harmless in fp16/fp32.  The first tile always
        // takes this path (m starts at 0, the first row max may be far below it).
        // (the vote needs no cross-lane exchange -- `any lane above` is the same before and after it: the LDS round trip of the
        // exchange is paid only where the shift is applied)
        if (t == 0 || __any(mx > RESCALE_THR)) {
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));  // max of S' = how far this tile exceeds the running max; both lanes of a query agree
            float delta = (t == 0) ? mx : fmaxf(mx, 0.f);
            if (mx == -INFINITY) delta = 0.f;  // fully masked row so far: nothing to shift
            if (t != 0) {  // nothing accumulated yet on the first tile
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
        if constexpr (HAS_ONES) {
#pragma unroll
            for (int kt = 0; kt < KV / 32; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) pb[kt][r >> 3][r & 7] = (f16)__builtin_amdgcn_exp2f(st[kt][r]);
        } else {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 ps2 = {0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < KV / 32; ++kt)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    f32x2 e;
                    e.x = __builtin_amdgcn_exp2f(st[kt][r]);
                    e.y = __builtin_amdgcn_exp2f(st[kt][r + 1]);
                    ps2 += e;
                    pb[kt][r >> 3][r & 7] = (f16)e.x;
                    pb[kt][r >> 3][(r & 7) + 1] = (f16)e.y;
                }
            l_run += ps2.x + ps2.y;
        }
    };
    // O^T += V^T P^T ; k-slot (hi, jj) of step (kt, s) <-> key kt*32 + 16*s + 8*(jj>>2) + 4*hi + (jj&3):
    // the A operand is two transpose reads (keys +0..3 and +8..11 of this lane's key half) of row-major V
    auto pv = [&](const int slot) __attribute__((always_inline)) {
        const lds_h4_ptr vbase = (lds_h4_ptr)(size_t)(lds0 + (unsigned)(slot * TILE_BYTES) + v_lane);
#pragma unroll
        for (int kt = 0; kt < KV / 32; ++kt)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int i = 0; i < DB; ++i) {
                    // one base VGPR per tile + immediate offsets (s16x4 units: 8 bytes)
                    const int off8 = (i * KV * 64 + (kt * 32 + 16 * s) * 64) / 8;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(vbase + off8);
                    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(vbase + off8 + 64);
                    const f16x8 a = __builtin_bit_cast(f16x8, __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pb[kt][s], acc[i], 0, 0, 0);
                }
            }
    };

    // One K/V tile.  HAS_NEXT is a compile-time flag: the steady-state iterations fetch tile t+1 unconditionally and the
    // last tile is peeled.
    auto tile = [&](const int t, auto has_next) {
        constexpr bool HAS_NEXT = decltype(has_next)::value;
        const int cur = (NBUF == 2) ? (t & 1) : 0;
        if constexpr (HAS_NEXT && NBUF == 2) stage(t + 1, cur ^ 1);  // lands while this tile is being consumed
        qk(cur);
        softmax(t);
        pv(cur);
        if constexpr (HAS_NEXT) {
            if constexpr (NBUF == 1) {  // single buffer: refill only after every wave is done with the tile
                __syncthreads();
                stage(t + 1, 0);
            }
            wait_vmcnt<0>();
            plant_ones((NBUF == 2) ? (cur ^ 1) : 0);
            __syncthreads();
        }
    };
    for (int t = 0; t + 1 < ntiles; ++t) tile(t, std::true_type{});
    tile(ntiles - 1, std::false_type{});

    // ---- normalise and store: lane holds d = i*32 + (r&3) + 8*(r>>2) + 4*hi for its query
    float l_tot;
    if constexpr (HAS_ONES) {
        // accumulator row D of the last block: local row D % 32 = (r & 3) + 8 * (r >> 2) with hi = 0 -> r = D % 32 / 2 (8 -> 4, 16 -> 8)
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
                    *reinterpret_cast<uint2*>(op + (size_t)qrow * p.ldo + d0) = o.u;
                }
            }
    }
}

// ---- long self-attention, head dim 40 (64x64 / 96x96 latents: lib/model_zoo/attention.py:170-193 with N = 4096 / 9216).
// Round 6.  SQ counters of the serial loop above (profiles/r05_pmc_sq.txt, tools/probes/attn_pmc.sh) show a SIMD that ISSUES
// instructions 88 % of its cycles while its matrix pipe is 0.52 busy: per 64-key tile and wave 14 MFMAs ride on ~130 other
// instructions (VALU 77, LDS 24, SALU 26, LDS-DMA 2) at ~5 cycles of issue each, and an ablation of a software-pipelined
// variant put numbers on each class (MFMAs alone 121 us of 306; LDS reads +67, row max +60, tile boundary +59 of which the two
// DMA requests +33, exp2 +45).  So this kernel spends fewer instructions per MFMA:
//   * a wave owns 64 queries (two 32-row blocks): K / V fragments, LDS-DMA requests, barriers and loop control are shared by
//     twice the MFMAs (22 LDS reads per 28 MFMAs instead of per 14);
//   * the threshold vote of the deferred rescale is ONE max over both row blocks, with no cross-lane exchange (the exchange
//     and the exact per-block maxima live in the rare branch);
//   * K and V sit in LDS slot-major ([16-byte chunk][key]), so a tile is 10 DMA pieces instead of 15 (no padding chunks
//     travel) and the constant columns -- the 1.0 behind V's column D that makes the P.V MFMA produce the row sums, and a
//     1.0 behind K's column D -- are written ONCE per ring slot instead of planted per tile;
//   * the running max m of a query rides in Q's spare k-slot D (as -m in fp16, against that constant 1.0 of K): the MFMA
//     subtracts it, S' = s - m leaves the matrix core ready for exp2 and no VGPR holds -m as a C operand.  m is kept as the
//     fp16 value that is actually subtracted, so the algebra stays exact: shifts are differences of representable values.
// The loop is software-pipelined inside the wave at half-tile (32-key) steps -- step h carries Q.K^T of step h + 1, the
// exp2 / pack of step h and P.V of step h - 1, pinned into one interleaved stream with sched_group_barrier -- and the fragment
// registers alternate: V fragments load under the Q.K^T MFMAs and are consumed by P.V, K fragments of the next step load
// under P.V.  Ring of 4 tile slots (slot = tile & 3), one barrier per tile.
template <int D>
__global__ __launch_bounds__(512, 2) void attn_pipe_kernel(const AttnArgs p) {
    static_assert(D == 40, "built for head dim 40: 3 k-steps with a spare slot at D, 2 V panels with a spare column at D");
    constexpr int NWV = 8, RB = 2;          // waves per block, 32-row query blocks per wave
    constexpr int QB = 32 * RB * NWV;       // 512 queries per block
    constexpr int KS = 3, DB = 2;
    constexpr int NKC = D / 8;              // 5 data chunks per K row
    constexpr int K_BYTES = (NKC + 1) * 1024;            // [chunk 0..4 | constant chunk 5][64 keys][16 B]
    constexpr int V0_BYTES = 4096;                       // panel 0: [64 keys][32 halfs]
    constexpr int V1_STRIDE = 1024 + 64;                 // panel 1 chunks staggered by 64 B: the three chunk groups of a transpose read hit different banks
    constexpr int V1_BYTES = 3 * V1_STRIDE + 64;         // chunk 0 = d 32..39 (DMA), chunk 1 = {1, 0 x 7} (constant), chunk 2 = zeros (constant)
    constexpr int SLOT_BYTES = ((K_BYTES + V0_BYTES + V1_BYTES + 1023) / 1024) * 1024;   // 14 KiB
    constexpr int NPIECE = NKC + 4 + 1;     // DMA pieces per tile: 5 K chunks, 4 x 16 keys of panel 0, chunk 0 of panel 1
    constexpr int NSLOT = 4;                // ring: tile t - 1 still feeds P.V while t + 1 feeds Q.K^T and t + 2 is in flight
    static_assert(NSLOT * SLOT_BYTES <= 64 * 1024, "static LDS");
    __shared__ __attribute__((aligned(1024))) char lds_all[NSLOT * SLOT_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    int qb, bh;
    {
        const int bid = blockIdx.x;
        if ((p.BH & 7) == 0) {   // an XCD owns BH / 8 consecutive (batch, head) pairs: K / V of a pair stay in one L2
            const int xcd = bid & 7, idx = bid >> 3;
            bh = xcd * (p.BH >> 3) + idx / p.nqb;
            qb = idx % p.nqb;
        } else {
            bh = bid / p.nqb;
            qb = bid % p.nqb;
        }
    }
    const int b = bh / p.H, h = bh % p.H;
    const f16* qp = p.q + (size_t)b * p.sq + h * D;
    const f16* kp = p.k + (size_t)b * p.sk + h * D;
    const f16* vp = p.v + (size_t)b * p.sv + h * D;
    f16* op = p.o + (size_t)b * p.so + h * D;

    // ---- staging plan: piece j of a tile is 1 KiB of LDS = lane * 16 bytes (buffer_load ... lds); rows past Nk fall outside
    // the descriptor and read as zeros.  Wave w requests piece w, waves 0 / 1 also pieces 8 / 9.
    const i32x4 rs_k = make_rsrc_words(kp, (unsigned)(((size_t)(p.Nk - 1) * p.ldk + D) * 2));
    const i32x4 rs_v = make_rsrc_words(vp, (unsigned)(((size_t)(p.Nk - 1) * p.ldv + D) * 2));
    auto piece_voff = [&](const int j) -> unsigned {
        if (j < NKC) return (unsigned)((lane * p.ldk + j * 8) * 2);                                              // K chunk j: key = lane
        if (j < NKC + 4) return (unsigned)((((j - NKC) * 16 + (lane >> 2)) * p.ldv + (lane & 3) * 8) * 2);       // panel 0, keys 16 (j - 5) ..
        return (unsigned)((lane * p.ldv + 32) * 2);                                                              // panel 1 chunk 0: key = lane
    };
    auto piece_dst = [&](const int j) -> unsigned {
        return j < NKC ? (unsigned)(j * 1024) : (j < NKC + 4 ? (unsigned)(K_BYTES + (j - NKC) * 1024) : (unsigned)(K_BYTES + V0_BYTES));
    };
    const bool a_is_k = wave < NKC;
    const i32x4 rs_a = a_is_k ? rs_k : rs_v;
    unsigned voff_a = piece_voff(wave), voff_b = piece_voff(8 + (wave & 1));
    const unsigned dst_a = piece_dst(wave), dst_b = piece_dst(8 + (wave & 1));
    const unsigned step_a = (unsigned)(KV * (a_is_k ? p.ldk : p.ldv) * 2), step_b = (unsigned)(KV * p.ldv * 2);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds_all;
    auto stage = [&](const int slot) {   // requests the NEXT tile in sequence (offsets advance by one tile per call)
        const unsigned base = lds0 + (unsigned)(slot * SLOT_BYTES);
        dma16(rs_a, base + dst_a, voff_a, 0);
        voff_a += step_a;
        if (wave < NPIECE - 8) {
            dma16(rs_v, base + dst_b, voff_b, 0);
            voff_b += step_b;
        }
    };
    // ---- constants of the ring slots (the DMA never touches them): K chunk 5 and V panel-1 chunk 1 hold {1, 0 x 7}
    // per key, V panel-1 chunk 2 zeros
    for (int i = tid; i < NSLOT * 3 * 64; i += 64 * NWV) {
        const int slot = i / 192, r = i % 192, region = r / 64, key = r % 64;
        const int off = region == 0 ? NKC * 1024 : K_BYTES + V0_BYTES + region * V1_STRIDE;
        *reinterpret_cast<uint4*>(lds_all + slot * SLOT_BYTES + off + key * 16) = make_uint4(region == 2 ? 0u : 0x3C00u, 0u, 0u, 0u);
    }

    // ---- Q fragments (B operand: lane = (query l31, k-half hi), 8 consecutive d per k-step), softmax scale * log2(e) folded in
    int qrow[RB];
    f16x8 qf[RB][KS];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        qrow[rb] = qb * QB + wave * (32 * RB) + rb * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d0 = ks * 16 + hi * 8;
            U4H8 t;
            t.u = make_uint4(0, 0, 0, 0);
            if (qrow[rb] < p.Nq && d0 < D) t.u = *reinterpret_cast<const uint4*>(qp + (size_t)qrow[rb] * p.ldq + d0);
#pragma unroll
            for (int j = 0; j < 8; ++j) t.e[j] = (f16)((float)t.e[j] * p.scale_log2);
            qf[rb][ks] = t.h;
        }
    }
    f32x16 acc[RB][DB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int i = 0; i < DB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][i][r] = 0.f;
    float m_cur[RB] = {0.f, 0.f};   // running max of each query (log2 units), always an fp16-representable value

    const int ntiles = (p.Nk + KV - 1) / KV;
    // per-lane LDS byte offsets of the operand reads inside a slot
    const unsigned k_lane = (unsigned)(hi * 1024 + l31 * 16);                                    // + ks * 2048 + kt * 512
    const unsigned v0_lane = (unsigned)(K_BYTES + (4 * hi + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8);
    const int v1c = ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1);                                   // chunk this lane feeds into the transpose
    const unsigned v1_lane = (unsigned)(K_BYTES + V0_BYTES + (v1c > 2 ? 2 : v1c) * V1_STRIDE + (4 * hi + ((lane & 15) >> 2)) * 16 + (lane & 1) * 8);
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4* lds_h4_ptr;
    typedef __attribute__((address_space(3))) f16x8* lds_h8_ptr;

    f32x16 st[2][RB];        // scores of the step parities
    f16x8 pb[2][RB][2];      // probabilities of the step parities: [16-key half of the step]
    f16x8 kf[KS];            // K fragments of the next Q.K^T
    s16x4 vf[2][DB][2];      // V fragments of the next P.V: [16-key half][panel][keys +0..3 / +8..11]
    auto read_k = [&](const unsigned kaddr, const int kt) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kf[ks] = *(lds_h8_ptr)(size_t)(kaddr + (unsigned)(ks * 2048 + kt * 512));
    };
    auto read_v = [&](const unsigned v0addr, const unsigned v1addr, const int kt) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const lds_h4_ptr b0 = (lds_h4_ptr)(size_t)(v0addr + (unsigned)((kt * 32 + 16 * s) * 64));
            const lds_h4_ptr b1 = (lds_h4_ptr)(size_t)(v1addr + (unsigned)((kt * 32 + 16 * s) * 16));
            vf[s][0][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(b0);
            vf[s][0][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(b0 + 64);   // + 8 keys (s16x4 units: 8 bytes)
            vf[s][1][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(b1);
            vf[s][1][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(b1 + 16);
        }
    };
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // S'^T (keys x queries) of both row blocks: consecutive MFMAs alternate between the two independent chains
    auto qk_chain = [&](f32x16 (&dst)[RB]) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) dst[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks], qf[rb][ks], ks == 0 ? zero16 : dst[rb], 0, 0, 0);
    };
    // O^T += V^T P^T
    auto pv_frags = [&](const f16x8 (&pp)[RB][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < DB; ++i) {
                const f16x8 a = __builtin_bit_cast(f16x8, __builtin_shufflevector(vf[s][i][0], vf[s][i][1], 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) acc[rb][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pp[rb][s], acc[rb][i], 0, 0, 0);
            }
    };
    auto mask_keys = [&](f32x16& sc, const int key0) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (key0 + (r & 3) + 8 * (r >> 2) + 4 * hi >= p.Nk) sc[r] = -INFINITY;
    };
    auto row_max = [&](const f32x16& sc) __attribute__((always_inline)) {   // two chains of v_max3_f32
        float m0 = fmaxf(fmaxf(sc[0], sc[1]), sc[2]), m1 = fmaxf(fmaxf(sc[8], sc[9]), sc[10]);
        m0 = fmaxf(fmaxf(m0, sc[3]), sc[4]);
        m1 = fmaxf(fmaxf(m1, sc[11]), sc[12]);
        m0 = fmaxf(fmaxf(m0, sc[5]), sc[6]);
        m1 = fmaxf(fmaxf(m1, sc[13]), sc[14]);
        return fmaxf(fmaxf(m0, m1), fmaxf(sc[7], sc[15]));
    };
    // shift the running max of row block rb by (about) delta >= 0: what moves is the fp16 value in Q's spare slot, and the
    // shift applied everywhere else is the exact difference of the two representable values
    auto shift_max = [&](const int rb, const float delta) __attribute__((always_inline)) -> float {
        const float tgt = m_cur[rb] + delta;
        f16 mh = (f16)tgt;
        if ((float)mh < tgt) {   // round UP: the applied shift is never smaller than the one asked for (P stays below 2^THR)
            unsigned short bits = __builtin_bit_cast(unsigned short, mh);
            bits = (unsigned short)((bits & 0x8000u) ? bits - 1 : bits + 1);
            mh = __builtin_bit_cast(f16, bits);
        }
        const float de = (float)mh - m_cur[rb];
        m_cur[rb] = (float)mh;
        if (hi) qf[rb][KS - 1][0] = -mh;   // k-slot D lives in the hi half of the last k-step
        return de;
    };

    // One step (32 keys); h = 2 t + P.  MODE 0: steady state; 1: the scores produced here belong to the last tile (key mask);
    // 2: last step (no Q.K^T, no vote)
    auto step = [&](auto par, auto mode, const int t) __attribute__((always_inline)) {
        constexpr int P = decltype(par)::value;
        constexpr int MODE = decltype(mode)::value;
        if constexpr (P == 0) {
            // tile boundary: tile t + 1 has landed (requested a whole tile ago) and becomes visible; every wave is done with
            // tile t - 2 (its last reads, V of step 2t - 3 ... 2t - 2, were waited for before this barrier) -> its slot takes tile t + 2
            wait_vmcnt<0>();
            __syncthreads();
            if (t + 2 < ntiles) stage((t + 2) & 3);
        }
        // V of step h - 1: tile t - 1 second half (P = 0) or tile t first half (P = 1); the very first step multiplies it by
        // P = 0, so any landed tile will do.  K of step h + 2: tile t + 1, half P.
        const int vslot = (P == 0 && t > 0) ? ((t - 1) & 3) : (t & 3);
        // addresses as finished VGPRs: address arithmetic inside the pinned region would sit between the instruction groups
        unsigned kaddr = lds0 + (unsigned)(((t + 1) & 3) * SLOT_BYTES) + k_lane;
        unsigned v0addr = lds0 + (unsigned)(vslot * SLOT_BYTES) + v0_lane, v1addr = lds0 + (unsigned)(vslot * SLOT_BYTES) + v1_lane;
        asm volatile("" : "+v"(kaddr), "+v"(v0addr), "+v"(v1addr));
        __builtin_amdgcn_sched_barrier(0);
        // ---- matrix stream: Q.K^T of step h + 1 on the K fragments in registers, V fragments of step h - 1 load under it;
        //      then P.V of step h - 1, the K fragments of step h + 2 load under it
        if constexpr (MODE != 2) qk_chain(st[P ^ 1]);
        read_v(v0addr, v1addr, P ^ 1);
        pv_frags(pb[P ^ 1]);
        if constexpr (MODE != 2) read_k(kaddr, P);
        // ---- VALU stream: P(h) = exp2(S'(h)) packed to fp16, then ONE max over both row blocks of S'(h + 1) for the vote
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) pb[P][rb][r >> 3][r & 7] = (f16)__builtin_amdgcn_exp2f(st[P][rb][r]);
        float mx = 0.f;
        if constexpr (MODE != 2) {
            if constexpr (MODE == 1) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) mask_keys(st[P ^ 1][rb], (2 * t + P + 1) * 32);
            }
            mx = fmaxf(row_max(st[P ^ 1][0]), row_max(st[P ^ 1][1]));
        }
        if constexpr (MODE == 0) {
            // 14 MFMAs; per gap: transcendentals (0x400), plain VALU (0x002: packs, then the max chains once Q.K^T is done), LDS reads (0x100)
#define VD_GAP(NT, NV, ND)                                                       \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                           \
    if (ND) __builtin_amdgcn_sched_group_barrier(0x100, ND, 0);                  \
    if (NT) __builtin_amdgcn_sched_group_barrier(0x400, NT, 0);                  \
    if (NV) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
            // (measured: this order 247 us; MFMAs in pairs 259-268; bare Q.K^T chain 263; the compiler's own order 257-262)
            VD_GAP(3, 1, 4) VD_GAP(3, 1, 4) VD_GAP(3, 1, 0) VD_GAP(3, 1, 0) VD_GAP(3, 1, 0) VD_GAP(3, 1, 0)
            VD_GAP(2, 2, 1) VD_GAP(2, 2, 1) VD_GAP(2, 4, 1) VD_GAP(2, 4, 0) VD_GAP(2, 4, 0) VD_GAP(2, 4, 0) VD_GAP(2, 4, 0) VD_GAP(0, 4, 0)
#undef VD_GAP
        }
        if constexpr (MODE != 2) {
            // the threshold test needs no cross-lane exchange: `any lane above` is the same vote before and after it
            if (__any(mx > RESCALE_THR)) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    float mr = row_max(st[P ^ 1][rb]);
                    mr = fmaxf(mr, __shfl_xor(mr, 32, 64));   // both lanes of a query agree on the shift
                    const float de = shift_max(rb, fmaxf(mr, 0.f));
                    const float alpha = __builtin_amdgcn_exp2f(-de);
                    const f16 ah = (f16)alpha;
#pragma unroll
                    for (int i = 0; i < DB; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[rb][i][r] *= alpha;
                    // P(h) is relative to the old m and not accumulated yet; S'(h + 1) was computed against the old m
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                        for (int j = 0; j < 8; ++j) pb[P][rb][s2][j] *= ah;
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[P ^ 1][rb][r] -= de;
                }
            }
        } else {
            asm volatile("" ::"v"(pb[P][0][0]), "v"(pb[P][0][1]), "v"(pb[P][1][0]), "v"(pb[P][1][1]));
        }
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using M0 = std::integral_constant<int, 0>;
    using M1 = std::integral_constant<int, 1>;
    using M2 = std::integral_constant<int, 2>;
    stage(0);
    // The Q loads are the only VMEM results the compiler tracks: consume them here so its s_waitcnt vmcnt(0) lands before the loop
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" ::"v"(qf[rb][ks]));
    wait_vmcnt<0>();
    __syncthreads();   // tile 0 and the slot constants are visible
    if (ntiles > 1) stage(1);
    // step "-1": scores of step 0 against m = 0, then the first row max sets m; nothing to accumulate yet (P = 0)
    read_k(lds0 + k_lane, 0);
    qk_chain(st[0]);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        mask_keys(st[0][rb], 0);
        float mr = row_max(st[0][rb]);
        mr = fmaxf(mr, __shfl_xor(mr, 32, 64));
        const float de = shift_max(rb, (mr == -INFINITY) ? 0.f : mr);
#pragma unroll
        for (int r = 0; r < 16; ++r) st[0][rb][r] -= de;
#pragma unroll
        for (int s = 0; s < 2; ++s) pb[1][rb][s] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    read_k(lds0 + k_lane, 1);
    int t = 0;
    for (; t + 2 < ntiles; ++t) {
        step(P0{}, M0{}, t);
        step(P1{}, M0{}, t);
    }
    if (ntiles >= 2) {
        step(P0{}, M0{}, t);
        step(P1{}, M1{}, t);
        ++t;
    }
    step(P0{}, M1{}, t);
    step(P1{}, M2{}, t);
    read_v(lds0 + (unsigned)((t & 3) * SLOT_BYTES) + v0_lane, lds0 + (unsigned)((t & 3) * SLOT_BYTES) + v1_lane, 1);
    pv_frags(pb[1]);

    // ---- normalise and store: lane holds d = i*32 + (r&3) + 8*(r>>2) + 4*hi for its query; the row sum sits in accumulator
    // row D of panel 1 (local row D % 32 = 8 -> register 4 of the hi = 0 lane)
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const float l_tot = __shfl(acc[rb][DB - 1][4], l31, 64);
        const float inv = (l_tot > 0.f) ? 1.0f / l_tot : 0.f;
        if (qrow[rb] < p.Nq) {
#pragma unroll
            for (int i = 0; i < DB; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d0 = i * 32 + 8 * g + 4 * hi;
                    if (d0 < D) {
                        U2H4 o;
#pragma unroll
                        for (int j = 0; j < 4; ++j) o.e[j] = (f16)(acc[rb][i][g * 4 + j] * inv);
                        *reinterpret_cast<uint2*>(op + (size_t)qrow[rb] * p.ldo + d0) = o.u;
                    }
                }
        }
    }
}

template <int D>
int launch_attn_pipe(AttnArgs a, hipStream_t stream) {
    a.nqb = (a.Nq + 511) / 512;
    hipLaunchKernelGGL(attn_pipe_kernel<D>, dim3(a.nqb * a.BH), dim3(512), 0, stream, a);
    return vd_check_launch("vd_attention_f16");
}

// ---- single-head attention over a WIDE head (D = 128 / 256 / 512): the AutoencoderKL mid-block AttnBlock
// (/root/reference/lib/model_zoo/autokl_modules.py:150-202: q, k, v 1x1 convs over C = 512 channels, torch.bmm -> softmax ->
// torch.bmm; 4096 tokens at 512x512, 9216 at 768x768).  The O accumulator of a 32-query block is 32 x 512 fp32 -- too wide for
// one wave -- so the head dim is SPLIT OVER THE FOUR WAVES of a block: wave w owns d in [w D/4, (w + 1) D/4).  Per tile of 32
// keys every wave computes the partial S^T = K[:, slice] Q^T[slice] of its slice, the four partials meet in LDS (one
// exchange, double-buffered), every wave then holds the full score tile, runs the same online softmax (fp32 statistics,
// deferred rescale) and accumulates O^T[slice] += V^T[slice] P^T.  No [N, N] tensor exists anywhere.
//   LDS per stage: K image [32 keys][D] (16-byte slots XOR-ed with key & 15 on the DMA source side: conflict-free
//   ds_read_b128 of 32 key rows), V image [D / 32 panels][32 keys][32] read transposed (ds_read_b64_tr_b16) like above;
//   two stages + two exchange buffers = 160 KiB at D = 512.
template <int D>
__global__ __launch_bounds__(256, 1) void attn_wide_kernel(const AttnArgs p) {
    constexpr int KVW = 32;                 // keys per tile
    constexpr int DW = D / 4;               // head-dim slice of a wave
    constexpr int KSW = DW / 16;            // k-steps of the partial QK^T
    constexpr int DBW = DW / 32;            // 32-row blocks of O^T per wave == V panels per wave
    constexpr int SPR = D / 8;              // 16-byte slots per K row
    constexpr int K_BYTES = KVW * D * 2;
    constexpr int V_BYTES = KVW * D * 2;
    constexpr int TILE_BYTES = K_BYTES + V_BYTES;
    constexpr int XCH_BYTES = 4 * 64 * 16 * 4;   // four partial score tiles: [wave][4 groups][64 lanes] float4
    constexpr int NPIECE = K_BYTES / 1024;       // DMA pieces per operand and tile
    constexpr int PPW = NPIECE / 4;              // ... per wave
    static_assert(D % 128 == 0 && D <= 512 && SPR >= 16 && NPIECE % 4 == 0, "head dim");
    extern __shared__ __attribute__((aligned(1024))) char wsm[];   // [2 stages][K | V] then [2][exchange]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.x / p.nqb, qb = blockIdx.x % p.nqb;
    const f16* qp = p.q + (size_t)b * p.sq;
    const f16* kp = p.k + (size_t)b * p.sk;
    const f16* vp = p.v + (size_t)b * p.sv;
    f16* op = p.o + (size_t)b * p.so;
    const i32x4 rs_k = make_rsrc_words(kp, (unsigned)(((size_t)(p.Nk - 1) * p.ldk + D) * 2));
    const i32x4 rs_v = make_rsrc_words(vp, (unsigned)(((size_t)(p.Nk - 1) * p.ldv + D) * 2));
    constexpr unsigned OOB = 0x80000000u;
    // K piece q = wave + 4 m: lane -> key (q * 64 + lane) / SPR, physical slot (q * 64 + lane) % SPR; fetches the logical slot
    // that lives there.  V piece q: chunk c = q * 64 + lane -> panel c >> 7, key (c >> 2) & 31, 8 channels (c & 3) of the panel.
    unsigned voff_k[PPW], voff_v[PPW];
#pragma unroll
    for (int m = 0; m < PPW; ++m) {
        const int c = (wave + 4 * m) * 64 + lane;
        const int key = c / SPR, ps = c % SPR;
        voff_k[m] = (unsigned)((key * p.ldk + ((ps ^ (key & 15)) << 3)) * 2);
        voff_v[m] = (unsigned)((((c >> 2) & 31) * p.ldv + (c >> 7) * 32 + (c & 3) * 8) * 2);
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)wsm;
    const unsigned k_tile_stride = (unsigned)(KVW * p.ldk * 2), v_tile_stride = (unsigned)(KVW * p.ldv * 2);
    auto stage = [&](int t, int buf) {
        const unsigned dst = lds0 + (unsigned)(buf * TILE_BYTES) + (unsigned)(wave * 1024);
        const unsigned kt_off = (unsigned)t * k_tile_stride, vt_off = (unsigned)t * v_tile_stride;
#pragma unroll
        for (int m = 0; m < PPW; ++m) dma16(rs_k, dst + m * 4096, voff_k[m] + kt_off, 0);
#pragma unroll
        for (int m = 0; m < PPW; ++m) dma16(rs_v, dst + K_BYTES + m * 4096, voff_v[m] + vt_off, 0);
    };

    // Q fragments of this wave's slice (B operand: lane = query l31, k-half hi), softmax scale * log2(e) folded in
    const int qrow = qb * 32 + l31;
    f16x8 qf[KSW];
#pragma unroll
    for (int ks = 0; ks < KSW; ++ks) {
        U4H8 t;
        t.u = make_uint4(0, 0, 0, 0);
        if (qrow < p.Nq) t.u = *reinterpret_cast<const uint4*>(qp + (size_t)qrow * p.ldq + wave * DW + ks * 16 + hi * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) t.e[j] = (f16)((float)t.e[j] * p.scale_log2);
        qf[ks] = t.h;
    }
    f32x16 acc[DBW];
#pragma unroll
    for (int i = 0; i < DBW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float m_run = 0.f, l_run = 0.f;   // running max (log2 units) and row sum of this lane's query (its key half)

    const int ntiles = (p.Nk + KVW - 1) / KVW;
    // K fragment of k-step ks: key row l31, logical slot wave * DW / 8 + 2 ks + hi
    int rd_k[KSW];
#pragma unroll
    for (int ks = 0; ks < KSW; ++ks) rd_k[ks] = l31 * (D * 2) + (((wave * (DW / 8) + 2 * ks + hi) ^ (l31 & 15)) << 4);
    const unsigned v_lane = (unsigned)(K_BYTES + (4 * hi + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8);
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4* lds_h4_ptr;

    stage(0, 0);
#pragma unroll
    for (int ks = 0; ks < KSW; ++ks) asm volatile("" ::"v"(qf[ks]));   // the compiler's own vmcnt(0) for the Q loads lands here
    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        wait_vmcnt<0>();
        __syncthreads();   // tile t has landed for every wave; every wave has left tile t - 1 (its stage is refilled below)
        if (t + 1 < ntiles) stage(t + 1, cur ^ 1);
        const char* Ks = wsm + cur * TILE_BYTES;
        // ---- partial S^T of this wave's slice
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks) {
            U4H8 a;
            a.u = *reinterpret_cast<const uint4*>(Ks + rd_k[ks]);
            st = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, qf[ks], st, 0, 0, 0);
        }
        // ---- exchange: [wave][group g][lane] float4, double-buffered over tiles (one barrier per tile)
        float* xw = reinterpret_cast<float*>(wsm + 2 * TILE_BYTES + cur * XCH_BYTES);
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(xw + ((wave * 4 + g) * 64 + lane) * 4) = make_float4(st[4 * g], st[4 * g + 1], st[4 * g + 2], st[4 * g + 3]);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 v = *reinterpret_cast<const float4*>(xw + ((w2 * 4 + g) * 64 + lane) * 4);
                st[4 * g] += v.x; st[4 * g + 1] += v.y; st[4 * g + 2] += v.z; st[4 * g + 3] += v.w;
            }
        // ---- online softmax for query `qrow`; this lane sees keys key0 + (r & 3) + 8 (r >> 2) + 4 hi
        const int key0 = t * KVW;
        if (key0 + KVW > p.Nk) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (key0 + (r & 3) + 8 * (r >> 2) + 4 * hi >= p.Nk) st[r] = -INFINITY;
        }
        float mx = st[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, st[r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        if (t == 0 || __any(mx - m_run > RESCALE_THR)) {
            const float m_new = (t == 0) ? mx : fmaxf(mx, m_run);
            if (t != 0) {
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                l_run *= alpha;
#pragma unroll
                for (int i = 0; i < DBW; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][r] *= alpha;
            }
            m_run = m_new;
        }
        f16x8 pb[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(st[r] - m_run);
            l_run += e;
            pb[r >> 3][r & 7] = (f16)e;
        }
        // ---- O^T[slice] += V^T[slice] P^T: panels DBW * wave .. of the V image, transposed on the way in
        const lds_h4_ptr vbase = (lds_h4_ptr)(size_t)(lds0 + (unsigned)(cur * TILE_BYTES) + v_lane);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int i = 0; i < DBW; ++i) {
                const int off8 = ((wave * DBW + i) * KVW * 64 + 16 * s2 * 64) / 8;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(vbase + off8);
                const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(vbase + off8 + 64);
                const f16x8 a = __builtin_bit_cast(f16x8, __builtin_shufflevector(lo, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pb[s2], acc[i], 0, 0, 0);
            }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = (l_tot > 0.f) ? 1.0f / l_tot : 0.f;
    if (qrow < p.Nq) {
#pragma unroll
        for (int i = 0; i < DBW; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = (wave * DBW + i) * 32 + 8 * g + 4 * hi;
                U2H4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o.e[j] = (f16)(acc[i][g * 4 + j] * inv);
                *reinterpret_cast<uint2*>(op + (size_t)qrow * p.ldo + d0) = o.u;
            }
    }
}

template <int D>
int launch_attn_wide(AttnArgs a, hipStream_t stream) {
    constexpr int LDS = 2 * (2 * 32 * D * 2) + 2 * (4 * 64 * 16 * 4);
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_wide_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) {
            vd_set_error("vd_attention_f16: cannot reserve %d bytes of LDS: %s", LDS, hipGetErrorString(e));
            return VD_ERR_LAUNCH;
        }
        done.fetch_or(bit, std::memory_order_release);
    }
    a.nqb = (a.Nq + 31) / 32;
    hipLaunchKernelGGL(attn_wide_kernel<D>, dim3(a.nqb * a.BH), dim3(256), LDS, stream, a);
    return vd_check_launch("vd_attention_f16");
}

template <typename OUT>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* s, OUT* pout, int n, float scale) {
    const size_t row = blockIdx.x;
    const float* sr = s + row * n;
    OUT* pr = pout + row * n;
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float mx = -INFINITY;
    for (int i = tid; i < n; i += 256) mx = fmaxf(mx, sr[i] * scale);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int i = tid; i < n; i += 256) sum += __expf(sr[i] * scale - mx);
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    for (int i = tid; i < n; i += 256) pr[i] = (OUT)(__expf(sr[i] * scale - mx) * inv);
}

template <int D, int NWV = 4>
int launch_attn(AttnArgs a, hipStream_t stream) {
    constexpr int QB = 32 * NWV;
    a.nqb = (a.Nq + QB - 1) / QB;
    a.ctx_map = (a.ctx_map && ((a.BH / a.H * a.nqb) & 7) == 0) ? 1 : 0;
    hipLaunchKernelGGL((attn_fwd_kernel<D, NWV>), dim3(a.nqb * a.BH), dim3(64 * NWV), 0, stream, a);
    return vd_check_launch("vd_attention_f16");
}

}  // namespace

extern "C" int vd_attention_f16(const void* q, const void* k, const void* v, void* out, int B, int H, int Nq, int Nk,
                                int D, int ldq, int ldk, int ldv, int ldo, int64_t sq, int64_t sk, int64_t sv,
                                int64_t so, float scale, int causal, hipStream_t stream) {
    VD_REQUIRE(q && k && v && out, "vd_attention_f16: null pointer");
    VD_REQUIRE(B > 0 && H > 0 && Nq > 0 && Nk > 0, "vd_attention_f16: empty problem B=%d H=%d Nq=%d Nk=%d", B, H, Nq, Nk);
    VD_REQUIRE((ldq % 8 == 0) && (ldk % 8 == 0) && (ldv % 8 == 0) && (ldo % 4 == 0),
               "vd_attention_f16: leading dimensions must keep 16-byte row alignment");
    VD_REQUIRE(((int64_t)Nk + KV) * ldk * 2 < (int64_t)1 << 31 && ((int64_t)Nk + KV) * ldv * 2 < (int64_t)1 << 31,
               "vd_attention_f16: one (batch, head) K/V slice must stay below 2 GiB (Nk=%d ldk=%d ldv=%d)", Nk, ldk, ldv);
    AttnArgs a;
    a.q = (const f16*)q; a.k = (const f16*)k; a.v = (const f16*)v; a.o = (f16*)out;
    a.H = H; a.Nq = Nq; a.Nk = Nk; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
    a.sq = sq; a.sk = sk; a.sv = sv; a.so = so;
    a.scale_log2 = scale * 1.44269504088896340736f;
    a.causal = causal;
    a.nqb = 0;   // set by the launcher (queries per block depend on the instantiation)
    a.BH = B * H;
    a.ctx_map = (Nk <= 2 * KV) ? 1 : 0;
    a.pair_contig = 1;
    const bool w8 = causal == 0 && Nq >= 2048 && Nk >= 1024;
    if (H == 1 && (D == 128 || D == 256 || D == 512)) {   // one wide head: head dim split over the waves of a block
        VD_REQUIRE(causal == 0, "vd_attention_f16: the wide single-head kernel has no causal mask");
        if (D == 128) return launch_attn_wide<128>(a, stream);
        if (D == 256) return launch_attn_wide<256>(a, stream);
        return launch_attn_wide<512>(a, stream);
    }
    switch (D) {
        case 40: {
            // long self-attention (64x64 / 96x96 latents): 64 queries per wave, software-pipelined (attn_pipe_kernel);
            // VD_ATTN_PIPE=0: the serial 8-wave loop
            static const char* pp_env = getenv("VD_ATTN_PIPE");
            if (w8) return (pp_env && pp_env[0] == '0') ? launch_attn<40, 8>(a, stream) : launch_attn_pipe<40>(a, stream);
            return launch_attn<40>(a, stream);
        }
        case 64: return launch_attn<64>(a, stream);
        case 80: return launch_attn<80>(a, stream);
        case 160: return launch_attn<160>(a, stream);
        default:
            vd_set_error("vd_attention_f16: unsupported head dim %d (supported: 40, 64, 80, 160; one head of 128 / 256 / 512)", D);
            return VD_ERR_UNSUPPORTED;
    }
}

extern "C" int vd_softmax_rows_f32_f16(const float* s, void* p, int64_t rows, int n, hipStream_t stream) {
    VD_REQUIRE(s && p && rows > 0 && n > 0, "vd_softmax_rows_f32_f16: bad arguments");
    VD_REQUIRE(rows < (1ll << 31), "vd_softmax_rows_f32_f16: too many rows");
    hipLaunchKernelGGL(softmax_rows_kernel<f16>, dim3((unsigned)rows), dim3(256), 0, stream, s, (f16*)p, n, 1.0f);
    return vd_check_launch("vd_softmax_rows_f32_f16");
}

extern "C" int vd_softmax_rows_f32_f32(const float* s, float* p, int64_t rows, int n, float scale, hipStream_t stream) {
    VD_REQUIRE(s && p && rows > 0 && n > 0, "vd_softmax_rows_f32_f32: bad arguments");
    VD_REQUIRE(rows < (1ll << 31), "vd_softmax_rows_f32_f32: too many rows");
    hipLaunchKernelGGL(softmax_rows_kernel<float>, dim3((unsigned)rows), dim3(256), 0, stream, s, p, n, scale);
    return vd_check_launch("vd_softmax_rows_f32_f32");
}
