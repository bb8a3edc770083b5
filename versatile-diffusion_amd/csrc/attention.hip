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
#include "vd_common.h"
#include "../../include/vd_hip.h"

namespace {

#ifndef VD_ATTN_MINW
#define VD_ATTN_MINW 1
#endif
constexpr int KV = 64;    // keys per tile
constexpr int QB = 128;   // queries per block (4 waves x 32)
constexpr int VROW = KV + 4;  // V^T row stride in halfs: 136 bytes -> conflict-free ds_read_b64
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
};

template <int D>
__global__ __launch_bounds__(256, (D <= 64 ? VD_ATTN_MINW : 1)) void attn_fwd_kernel(const AttnArgs p) {
    constexpr int KS = (D + 15) / 16;       // k-steps of the QK^T MFMA
    constexpr int DB = (D + 31) / 32;       // 32-row blocks of O^T
    constexpr int KROW = KS * 16 + 8;       // K row stride in halfs (odd number of 16-byte slots)
    constexpr int KCH = KS * 2;             // 16-byte chunks per staged K row (zero padded past D)
    constexpr int VCH = D / 8;              // 16-byte chunks per V row
    constexpr int K_ITERS = (KV * KCH + 255) / 256;
    constexpr int VITEMS = (KV / 2) * VCH;  // one item = 2 adjacent keys x 8 channels -> 8 dword LDS writes
    constexpr int V_ITERS = (VITEMS + 255) / 256;
    static_assert(D % 8 == 0, "head dim must be a multiple of 8");
    // double-buffered K / V^T tiles (one barrier per tile) where two copies fit the 64 KiB static LDS limit
    constexpr int TILE_HALFS = KV * KROW + DB * 32 * VROW;
    constexpr int NBUF = (2 * TILE_HALFS * 2 <= 60 * 1024) ? 2 : 1;

    __shared__ __attribute__((aligned(16))) f16 lds_all[NBUF * TILE_HALFS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hi = lane >> 5, l31 = lane & 31;

    // XCD-aware mapping: keep all query blocks of one (batch, head) on one XCD so K/V stay in its L2
    int qb, bh;
    {
        const int bid = blockIdx.x;
        if ((p.BH & 7) == 0) {
            const int xcd = bid & 7, idx = bid >> 3;
            bh = xcd + 8 * (idx / p.nqb);
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

    // zero the tiles once (V^T rows d >= D must be finite)
    for (int i = tid; i < NBUF * TILE_HALFS / 2; i += 256) reinterpret_cast<uint32_t*>(lds_all)[i] = 0u;

    // ---- Q fragments: B operand, lane = (query l31, k-half hi), 8 consecutive d per k-step
    const int qrow = qb * QB + wave * 32 + l31;
    f16x8 qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int d0 = ks * 16 + hi * 8;
        U4H8 t;
        t.u = make_uint4(0, 0, 0, 0);
        if (qrow < p.Nq && d0 < D) t.u = *reinterpret_cast<const uint4*>(qp + (size_t)qrow * p.ldq + d0);
        qf[ks] = t.h;
    }

    f32x16 acc[DB];
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    int ntiles = (p.Nk + KV - 1) / KV;
    if (p.causal) {
        const int last_q = min(qb * QB + QB - 1, p.Nq - 1);
        ntiles = min(ntiles, last_q / KV + 1);
    }

    uint4 rk[K_ITERS], rv[V_ITERS][2];
    auto load_kv = [&](int t) {
        const int key0 = t * KV;
#pragma unroll
        for (int it = 0; it < K_ITERS; ++it) {
            const int c = tid + it * 256;
            const int r = c / KCH, ch = c - r * KCH;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (c < KV * KCH && key0 + r < p.Nk && ch * 8 < D)
                v = *reinterpret_cast<const uint4*>(kp + (size_t)(key0 + r) * p.ldk + ch * 8);
            rk[it] = v;
        }
#pragma unroll
        for (int it = 0; it < V_ITERS; ++it) {
            const int c = tid + it * 256;
            const int kpair = c % (KV / 2), ch = c / (KV / 2);  // key pair fastest: conflict-free transposed writes
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int key = key0 + 2 * kpair + h2;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (c < VITEMS && key < p.Nk) v = *reinterpret_cast<const uint4*>(vp + (size_t)key * p.ldv + ch * 8);
                rv[it][h2] = v;
            }
        }
    };
    auto store_kv = [&](int buf) {
        f16* Ks = lds_all + buf * TILE_HALFS;
        f16* Vt = Ks + KV * KROW;
#pragma unroll
        for (int it = 0; it < K_ITERS; ++it) {
            const int c = tid + it * 256;
            const int r = c / KCH, ch = c - r * KCH;
            if (c < KV * KCH) *reinterpret_cast<uint4*>(Ks + r * KROW + ch * 8) = rk[it];
        }
#pragma unroll
        for (int it = 0; it < V_ITERS; ++it) {
            const int c = tid + it * 256;
            const int kpair = c % (KV / 2), ch = c / (KV / 2);
            if (c < VITEMS) {
                U4H8 a, b;
                a.u = rv[it][0];
                b.u = rv[it][1];
#pragma unroll
                for (int i = 0; i < 8; ++i) {  // V^T[d][2*kpair .. +1] as one dword: lanes -> consecutive banks
                    f16x2 pr;
                    pr[0] = a.e[i];
                    pr[1] = b.e[i];
                    *reinterpret_cast<f16x2*>(Vt + (ch * 8 + i) * VROW + 2 * kpair) = pr;
                }
            }
        }
    };

    if (ntiles > 0) {
        load_kv(0);
        __syncthreads();  // orders the zero fill
        store_kv(0);
    }
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int cur = (NBUF == 2) ? (t & 1) : 0;
        const f16* Ks = lds_all + cur * TILE_HALFS;
        const f16* Vt = Ks + KV * KROW;
        if (t + 1 < ntiles) load_kv(t + 1);  // in flight under the MFMAs below

        // ---- S^T tiles (keys x queries)
        f32x16 st[KV / 32];
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < KV / 32; ++kt) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                U4H8 a;
                a.u = *reinterpret_cast<const uint4*>(Ks + (kt * 32 + l31) * KROW + ks * 16 + hi * 8);
                // first k-step takes C = 0 as an inline constant: no 16-register zero fill per tile
                st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, qf[ks], ks == 0 ? zero16 : st[kt], 0, 0, 0);
            }
        }

        // V^T fragments of this tile are requested BEFORE the softmax math so the LDS latency hides under the VALU
        // work instead of stalling each P.V MFMA (affordable for head dims <= 96: 16 / 24 fragment registers pairs)
        constexpr bool PREFETCH_V = false;  // measured: +14 VGPRs drop 3 -> 2 waves/SIMD and cost 13 %; occupancy wins
        uint2 vfrag[PREFETCH_V ? KV / 32 : 1][2][PREFETCH_V ? DB : 1][2];
        if constexpr (PREFETCH_V) {
#pragma unroll
            for (int kt = 0; kt < KV / 32; ++kt)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int i = 0; i < DB; ++i) {
                        const f16* vr = Vt + (i * 32 + l31) * VROW + kt * 32 + 16 * s2 + 4 * hi;
                        vfrag[kt][s2][i][0] = *reinterpret_cast<const uint2*>(vr);
                        vfrag[kt][s2][i][1] = *reinterpret_cast<const uint2*>(vr + 8);
                    }
        }

        // ---- online softmax for query `qrow`; this lane sees keys key0 + kt*32 + (r&3)+8*(r>>2)+4*hi
        const int key0 = t * KV;
        // masking is needed only on the ragged last tile / on tiles that cross the causal diagonal (wave-uniform)
        const bool need_mask = (key0 + KV > p.Nk) || (p.causal && (key0 + KV - 1 > qb * QB + wave * 32));
        if (need_mask) {
#pragma unroll
            for (int kt = 0; kt < KV / 32; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    if (key >= p.Nk || (p.causal && key > qrow)) st[kt][r] = -INFINITY;
                }
        }
        float mx = st[0][0];
#pragma unroll
        for (int kt = 0; kt < KV / 32; ++kt)
#pragma unroll
            for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, st[kt][r]), st[kt][r + 1]);  // v_max3_f32
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float ms = mx * p.scale_log2;  // running max is tracked in scaled (log2) units
        // Deferred rescale: the accumulators are only rescaled when some row's max grew by more than RESCALE_THR
        // (then every lane updates exactly); otherwise P = exp2(s - m_old) <= 2^RESCALE_THR, harmless in fp16/fp32.
        if (__any(ms > m_run + RESCALE_THR)) {
            const float m_new = fmaxf(m_run, ms);
            const float alpha = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < DB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] *= alpha;
            m_run = m_new;
        }
        const float neg_m = (m_run == -INFINITY) ? 0.f : -m_run;
        // the softmax body is VALU-issue bound (3 waves per SIMD share one VALU port): packed fp32 math for the
        // scale/shift and the row sum halves those instruction counts
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 sc2 = {p.scale_log2, p.scale_log2}, nm2 = {neg_m, neg_m};
        f32x2 ps2 = {0.f, 0.f};
        f16x8 pb[KV / 32][2];
#pragma unroll
        for (int kt = 0; kt < KV / 32; ++kt)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                f32x2 v = {st[kt][r], st[kt][r + 1]};
                v = __builtin_elementwise_fma(v, sc2, nm2);
                f32x2 e;
                e.x = __builtin_amdgcn_exp2f(v.x);
                e.y = __builtin_amdgcn_exp2f(v.y);
                ps2 += e;
                pb[kt][r >> 3][r & 7] = (f16)e.x;
                pb[kt][r >> 3][(r & 7) + 1] = (f16)e.y;
            }
        l_run += ps2.x + ps2.y;

        // ---- O^T += V^T P^T ; k-slot (hi, jj) of step (kt, s) <-> key kt*32 + 16*s + 8*(jj>>2) + 4*hi + (jj&3)
#pragma unroll
        for (int kt = 0; kt < KV / 32; ++kt)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int kb = kt * 32 + 16 * s + 4 * hi;
#pragma unroll
                for (int i = 0; i < DB; ++i) {
                    const f16* vr = Vt + (i * 32 + l31) * VROW + kb;
                    U2H4 lo, hi4;
                    if constexpr (PREFETCH_V) {
                        lo.u = vfrag[kt][s][i][0];
                        hi4.u = vfrag[kt][s][i][1];
                    } else {
                        lo.u = *reinterpret_cast<const uint2*>(vr);
                        hi4.u = *reinterpret_cast<const uint2*>(vr + 8);
                    }
                    f16x8 a;
                    a[0] = lo.e[0]; a[1] = lo.e[1]; a[2] = lo.e[2]; a[3] = lo.e[3];
                    a[4] = hi4.e[0]; a[5] = hi4.e[1]; a[6] = hi4.e[2]; a[7] = hi4.e[3];
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pb[kt][s], acc[i], 0, 0, 0);
                }
            }
        // stage the next tile: into the other buffer (one barrier per tile) or, single-buffered, after everyone is done
        if (NBUF == 1) __syncthreads();
        if (t + 1 < ntiles) store_kv((NBUF == 2) ? ((t + 1) & 1) : 0);
        __syncthreads();
    }

    // ---- normalise and store: lane holds d = i*32 + (r&3) + 8*(r>>2) + 4*hi for its query
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
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

__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* s, f16* pout, int n) {
    const size_t row = blockIdx.x;
    const float* sr = s + row * n;
    f16* pr = pout + row * n;
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float mx = -INFINITY;
    for (int i = tid; i < n; i += 256) mx = fmaxf(mx, sr[i]);
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float sum = 0.f;
    for (int i = tid; i < n; i += 256) sum += __expf(sr[i] - mx);
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    for (int i = tid; i < n; i += 256) pr[i] = (f16)(__expf(sr[i] - mx) * inv);
}

template <int D>
int launch_attn(const AttnArgs& a, hipStream_t stream) {
    hipLaunchKernelGGL(attn_fwd_kernel<D>, dim3(a.nqb * a.BH), dim3(256), 0, stream, a);
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
    AttnArgs a;
    a.q = (const f16*)q; a.k = (const f16*)k; a.v = (const f16*)v; a.o = (f16*)out;
    a.H = H; a.Nq = Nq; a.Nk = Nk; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
    a.sq = sq; a.sk = sk; a.sv = sv; a.so = so;
    a.scale_log2 = scale * 1.44269504088896340736f;
    a.causal = causal;
    a.nqb = (Nq + QB - 1) / QB;
    a.BH = B * H;
    switch (D) {
        case 40: return launch_attn<40>(a, stream);
        case 64: return launch_attn<64>(a, stream);
        case 80: return launch_attn<80>(a, stream);
        case 160: return launch_attn<160>(a, stream);
        default:
            vd_set_error("vd_attention_f16: unsupported head dim %d (supported: 40, 64, 80, 160)", D);
            return VD_ERR_UNSUPPORTED;
    }
}

extern "C" int vd_softmax_rows_f32_f16(const float* s, void* p, int64_t rows, int n, hipStream_t stream) {
    VD_REQUIRE(s && p && rows > 0 && n > 0, "vd_softmax_rows_f32_f16: bad arguments");
    VD_REQUIRE(rows < (1ll << 31), "vd_softmax_rows_f32_f16: too many rows");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, stream, s, (f16*)p, n);
    return vd_check_launch("vd_softmax_rows_f32_f16");
}
