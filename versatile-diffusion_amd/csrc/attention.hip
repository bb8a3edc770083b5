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
#ifndef VD_ATTN_STAG_MINW
#define VD_ATTN_STAG_MINW 4   // waves per SIMD the register budget is set for: two 8-wave blocks per CU (<= 128 VGPRs)
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
// STAG (8 waves, round 5): the two waves of a SIMD (w and w + 4) run HALF AN ITERATION APART -- while one half of the block is in
// its softmax (VALU), the other half is in P.V / Q.K^T (matrix pipe); see the loop at the end of the kernel.
template <int D, int NWV = 4, bool STAG = false>
__global__ __launch_bounds__(64 * NWV, (STAG ? VD_ATTN_STAG_MINW : (NWV == 8 ? VD_ATTN_W8_MINW : (D <= 64 && NWV == 4 ? VD_ATTN_MINW : 1)))) void attn_fwd_kernel(const AttnArgs p) {
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

    constexpr int NSLOT = STAG ? 4 : NBUF;    // STAG: ring of 4 tiles (t - 1 .. t + 2 are alive inside iteration t)
    static_assert(!STAG || (NWV == 8 && 4 * TILE_BYTES <= 64 * 1024), "the staggered loop is built for 8 waves and a 4-tile ring in static LDS");
    __shared__ __attribute__((aligned(1024))) char lds_all[NSLOT * TILE_BYTES];

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
            // two of them (counter traffic of the D = 40 launch: 2.1x the unique bytes).  VD_ATTN_PAIRMAP=0: the old interleave.
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

    if constexpr (!STAG) {
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
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));  // max of S' = how far this tile exceeds the running max
        // Deferred rescale: accumulators and running max move only when some row grew by more than RESCALE_THR (then every
        // lane updates exactly); otherwise P = exp2(S') <= 2^RESCALE_THR, harmless in fp16/fp32.  The first tile always
        // takes this path (m starts at 0, the first row max may be far below it).
        if (t == 0 || __any(mx > RESCALE_THR)) {
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

    if constexpr (STAG) {
        // Both halves run the same two phases per tile -- V: softmax(t) -> P;  M: O += P.V(t), S(t + 1) = Q.K(t + 1)^T -- but
        // waves 4-7 lag HALF an iteration behind waves 0-3, so between two barriers a SIMD sees
        //     wave w     (early):  softmax(t)          | P.V(t), Q.K(t + 1)
        //     wave w + 4 (late):   P.V(t - 1), Q.K(t)  | softmax(t)
        // i.e. one wave's VALU phase beside the other's matrix phase instead of both queueing on the same pipe (lock-step:
        // pipes 0.47 / 0.51 busy and never together, profiles/r05_pmc_sq.txt).  Alive inside iteration t: tiles t - 1 (late P.V),
        // t, t + 1 (early Q.K) and t + 2 (in flight) = the 4-slot ring.
        const bool late = wave >= NWV / 2;
        stage(0, 0);
        if (ntiles > 1) stage(1, 1);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" ::"v"(qf[ks]));
        wait_vmcnt<0>();
        plant_ones(0);
        if (ntiles > 1) plant_ones(1);
        __syncthreads();
        // two self-contained loops (same barrier count): one loop with the order chosen inside it made the compiler keep both
        // halves' live ranges at once (201 VGPRs)
        auto run = [&](auto is_late) __attribute__((always_inline)) {
            constexpr bool LATE = decltype(is_late)::value;
            if constexpr (!LATE) qk(0);
            for (int t = 0; t < ntiles; ++t) {
                if (t + 2 < ntiles) stage(t + 2, (t + 2) & 3);   // slot of tile t - 2: its last reader (late P.V) finished before the barrier
                if constexpr (LATE) {
                    if (t > 0) pv((t - 1) & 3);
                    qk(t & 3);
                    softmax(t);
                } else {
                    softmax(t);
                    pv(t & 3);
                    if (t + 1 < ntiles) qk((t + 1) & 3);
                }
                if (t + 1 < ntiles) {
                    wait_vmcnt<0>();
                    if (t + 2 < ntiles) plant_ones((t + 2) & 3);
                    __syncthreads();
                }
            }
            if constexpr (LATE) pv((ntiles - 1) & 3);
        };
        if (late) run(std::true_type{});
        else run(std::false_type{});
    } else {
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
    }

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

template <int D, int NWV = 4, bool STAG = false>
int launch_attn(AttnArgs a, hipStream_t stream) {
    constexpr int QB = 32 * NWV;
    a.nqb = (a.Nq + QB - 1) / QB;
    a.ctx_map = (a.ctx_map && ((a.BH / a.H * a.nqb) & 7) == 0) ? 1 : 0;
    hipLaunchKernelGGL((attn_fwd_kernel<D, NWV, STAG>), dim3(a.nqb * a.BH), dim3(64 * NWV), 0, stream, a);
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
    static const char* ctx_env = getenv("VD_ATTN_CTXMAP");   // development switch: 0 = always the K/V-locality mapping
    a.ctx_map = (Nk <= 2 * KV && !(ctx_env && ctx_env[0] == '0')) ? 1 : 0;
    static const char* pm_env = getenv("VD_ATTN_PAIRMAP");   // development switch: 0 = pairs interleaved over the XCDs (rounds 1-4)
    a.pair_contig = (pm_env && pm_env[0] == '0') ? 0 : 1;
    static const char* w8_env = getenv("VD_ATTN_W8");        // development switch: 0 = always 4 waves per block
    const bool w8 = !(w8_env && w8_env[0] == '0') && causal == 0 && Nq >= 2048 && Nk >= 1024;
    if (H == 1 && (D == 128 || D == 256 || D == 512)) {   // one wide head: head dim split over the waves of a block
        VD_REQUIRE(causal == 0, "vd_attention_f16: the wide single-head kernel has no causal mask");
        if (D == 128) return launch_attn_wide<128>(a, stream);
        if (D == 256) return launch_attn_wide<256>(a, stream);
        return launch_attn_wide<512>(a, stream);
    }
    switch (D) {
        case 40: {
            // opt-in (VD_ATTN_STAG=1): the half-blocks half an iteration apart.  Correct (test_attention_staggered_halves) and measured
            // SLOWER: 274-283 us against 265 isolated, forward unchanged; with one block per CU (pure pairing) 313 us -- the two
            // waves of a SIMD do not overlap one's softmax with the other's MFMAs any better than two lock-stepped blocks do
            static const char* st_env = getenv("VD_ATTN_STAG");
            if (w8) return (st_env && st_env[0] == '1') ? launch_attn<40, 8, true>(a, stream) : launch_attn<40, 8>(a, stream);
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
