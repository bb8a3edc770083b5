// fp16 MFMA GEMM / implicit-GEMM convolution for gfx950 (CDNA4).
//
//   out[m][n] = epilogue( sum_k A[m][k] * W[n][k] )
//
// A is never materialised for convolutions: the tile loader gathers the (kh,kw) tap of an
// NHWC activation (optionally the channel-concatenation of two tensors, optionally nearest-2x
// upsampled, optionally strided) straight into LDS.  W is K-contiguous ([N][K], K ordered
// (kh, kw, cin)), i.e. torch Linear weights as they are and conv weights repacked once at load.
//
// Replaces on the reference path (all stock torch ops there):
//   nn.Conv2d 3x3 / 1x1 in ResBlock, Up/Downsample, SpatialTransformer.proj_in/out
//     (/root/reference/lib/model_zoo/openaimodel.py:89-117,133-159,254-274,
//      /root/reference/lib/model_zoo/attention.py:255-266)
//   nn.Linear in CrossAttention / GEGLU FeedForward / time_embed / emb_layers
//     (/root/reference/lib/model_zoo/attention.py:37-64,170-193, openaimodel.py:2627-2633)
//   the VAE convs and AttnBlock bmm's (/root/reference/lib/model_zoo/autokl_modules.py:82-202)
//
// Structure: 256 threads = 4 waves, BMxBN block tile, BK = 64, register-staged global->LDS
// double buffer (one barrier per K tile), XOR-swizzled LDS rows (conflict-free ds_read_b128),
// v_mfma_f32_32x32x16_f16 with fp32 accumulation, epilogue staged through LDS so that global
// stores are full 16-byte row segments with bias / per-batch row vector / activation / scale /
// residual fused.  Optional split-K (fp32 slabs + reduce kernel) for the small-M levels.
#include "vd_common.h"
#include "../../include/vd_hip.h"
#include <stdlib.h>

namespace {

constexpr int VD_GEMM_DEFAULT_DMA = 2;  // LDS-DMA, two stages (measured: +8..15 % over register staging; 3 stages lose occupancy)
constexpr int VD_GEMM_DEFAULT_DEEP = 3;
constexpr int BK = 64;           // K tile (halfs); one LDS row = 128 bytes = 8 x 16-byte slots
constexpr int ROW_BYTES = BK * 2;

struct GemmArgs {
    VdGemmDesc d;
    int tiles_m, tiles_n, kt_total, kt_per_split;
    unsigned a0_bytes, a1_bytes, w_bytes;  // per-batch operand extents for the buffer descriptors (< 2^31)
    int plain;                             // 1x1, stride 1, no pad / upsample, output grid == input grid
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB_OFFSET = 0x80000000u;  // beyond every descriptor's num_records -> hardware returns zeros

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void vd_store16_nt(void* p, uint4 v) {
#ifdef VD_NO_NT_STORE
    *reinterpret_cast<uint4*>(p) = v;
#else
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<u32x4_t*>(p));
#endif
}
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// swizzled byte offset of (row r, 16-byte slot s) inside a [rows][64] f16 LDS tile
__device__ __forceinline__ int lds_off(int r, int s) { return r * ROW_BYTES + ((s ^ ((r >> 1) & 7)) << 4); }
// same idea for 32-deep K tiles (64-byte rows, 4 slots): rows r, r+4, r+8, r+12 of a ds_read_b128 lane group land on
// the same 64-byte quarter of the 256-byte bank row and are spread over its 4 slots by (r >> 2) & 3
template <int KB>
__device__ __forceinline__ int lds_off_kb(int r, int s) {
    if constexpr (KB == 64) return lds_off(r, s);
    else return r * 64 + ((s ^ ((r >> 2) & 3)) << 4);
}
template <int KB>
__device__ __forceinline__ int lds_swz(int r) { return KB == 64 ? ((r >> 1) & 7) : ((r >> 2) & 3); }

struct EpiCtx {
    const f16* bias;
    const f16* rowvec;
    const f16* res;
    void* out;
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
    e.N = (d.act == VD_ACT_GEGLU) ? d.N / 2 : d.N;
    e.ldc = d.ldc;
    e.ldr = d.ldr;
    e.rows_per_batch = d.rows_per_batch > 0 ? d.rows_per_batch : 1;
    e.flags = d.flags;
    e.act = d.act;
    e.alpha = d.alpha;
    return e;
}

__device__ __forceinline__ float apply_act(int act, float v) {
    if (act == VD_ACT_QUICK_GELU) return vd_quick_gelu(v);
    if (act == VD_ACT_SILU) return vd_silu(v);
    return v;
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

// Epilogue part-2 helpers; CH = 16-byte chunks per output row of the block tile (compile-time: the chunk -> (row, col)
// split is a shift, not an integer division).
template <int BM, int CH, int NT, int MAX_CH>
__device__ __forceinline__ void epi_prefetch(const EpiCtx& e, int M, int m0, int out_n0, int tid, uint4* pre_res, uint4* pre_rv) {
    const bool vec_ok = ((e.N & 7) == 0) && ((e.ldr & 7) == 0);
#pragma unroll
    for (int k = 0; k < MAX_CH; ++k) {
        pre_res[k] = make_uint4(0, 0, 0, 0);
        pre_rv[k] = make_uint4(0, 0, 0, 0);
        const int c = tid + k * NT;
        if (vec_ok && c < BM * CH) {
            const int r = c / CH, cc = (c % CH) * 8;
            const int row = m0 + r, col = out_n0 + cc;
            if (row < M && col + 8 <= e.N) {
                if (e.flags & VD_EPI_RESIDUAL) pre_res[k] = *reinterpret_cast<const uint4*>(e.res + (size_t)row * e.ldr + col);
                if (e.flags & VD_EPI_ROWVEC)
                    pre_rv[k] = *reinterpret_cast<const uint4*>(e.rowvec + (size_t)(row / e.rows_per_batch) * e.N + col);
            }
        }
    }
}

template <int BM, int CH, int NT, int MAX_CH, int CS_LD>
__device__ __forceinline__ void epi_writeout(const EpiCtx& e, int M, int m0, int out_n0, int tid, const f16* cs,
                                             const uint4* pre_res, const uint4* pre_rv) {
    const bool vec_ok = ((e.N & 7) == 0) && ((e.ldr & 7) == 0) && ((e.ldc & 7) == 0);
#pragma unroll
    for (int k = 0; k < MAX_CH; ++k) {
        const int c = tid + k * NT;
        if (c < BM * CH) {
            const int r = c / CH, cc = (c % CH) * 8;
            const int row = m0 + r, col = out_n0 + cc;
            if (row < M && col < e.N) {
                U4H8 t;
                t.u = *reinterpret_cast<const uint4*>(cs + r * CS_LD + cc);
                if (vec_ok && col + 8 <= e.N) {
                    U4H8 a, b, o;
                    a.u = pre_res[k];
                    b.u = pre_rv[k];
#pragma unroll
                    for (int i = 0; i < 8; ++i) o.e[i] = (f16)((float)t.e[i] + (float)a.e[i] + (float)b.e[i]);
                    // streaming output: non-temporal so 20..80 MB of results do not evict the weight / activation panels
                    // that the other tiles of this XCD keep re-reading from its 4 MiB L2
                    vd_store16_nt(reinterpret_cast<f16*>(e.out) + (size_t)row * e.ldc + col, o.u);
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

template <int BM, int BN, int WM, int WN, int NT, int STAGES, int KB = 64>
__global__ __launch_bounds__(NT, (NT == 512 && WM * WN < 64 * 64 ? 4 : 2)) void gemm_f16_kernel(const GemmArgs p) {
    static_assert(KB == 64 || (KB == 32 && STAGES >= 2), "32-deep K tiles exist for the LDS-DMA ring only");
    constexpr int KROW_BYTES = KB * 2;  // one LDS row of a stage
    constexpr int SLOTS = KB / 8;       // 16-byte slots per row == threads cooperating on a row
    constexpr int KSUB = 64 / KB;       // kernel K tiles per planner K tile (the planner counts in 64s)
    constexpr int WAVES_N = BN / WN;
    constexpr int WAVES_M = BM / WM;
    static_assert(WAVES_M * WAVES_N * 64 == NT, "waves must tile the block");
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int RPP = NT / SLOTS;  // rows staged per pass: SLOTS threads per row
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile must be a multiple of the staging pass");
    constexpr int A_PASSES = BM / RPP, B_PASSES = BN / RPP;
    constexpr int STAGE_BYTES = (BM + BN) * KROW_BYTES;
    constexpr int CS_LD = BN + 8;  // fp16 epilogue tile leading dimension (halfs); row stride = odd multiple of 16 B

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const VdGemmDesc& d = p.d;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int hi = lane >> 5, l31 = lane & 31;

    // ---- XCD-aware tile mapping: block b runs on XCD b%8; give each XCD a contiguous run of
    // logical tiles (n fastest) so neighbouring tiles that share the A row-panel share an L2.
    const int ntiles = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / p.tiles_n, tn = bid - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int split = blockIdx.y;
    const int z = blockIdx.z;

    // ---- operands are read through buffer descriptors: one SGPR base + per-lane 32-bit byte offset + a scalar
    // K offset per tile.  Out-of-image taps, rows >= M / N and the ragged K tail simply use an out-of-range
    // offset and the hardware returns zeros: no branches and (within a conv tap) no VALU work per K tile.
    const __amdgpu_buffer_rsrc_t rs_a0 = make_rsrc(reinterpret_cast<const f16*>(d.a0) + (size_t)z * d.stride_a, p.a0_bytes);
    const __amdgpu_buffer_rsrc_t rs_a1 = make_rsrc(d.a1 ? reinterpret_cast<const f16*>(d.a1) + (size_t)z * d.stride_a : d.a0,
                                                   d.a1 ? p.a1_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(reinterpret_cast<const f16*>(d.w) + (size_t)z * d.stride_w, p.w_bytes);

    // per-thread gather coordinates: 8 threads per 128-byte row, RPP rows per pass
    const int lrow = tid / SLOTS, lslot = tid % SLOTS;
    int a_iy0[A_PASSES], a_ix0[A_PASSES], a_pix[A_PASSES];
    const int HWo = d.Hout * d.Wout;
    const int Hv = d.Hin << d.ups, Wv = d.Win << d.ups;
#pragma unroll
    for (int ps = 0; ps < A_PASSES; ++ps) {
        const int m = m0 + lrow + RPP * ps;
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
    unsigned voff_b[B_PASSES];
#pragma unroll
    for (int ps = 0; ps < B_PASSES; ++ps) {
        const int n = n0 + lrow + RPP * ps;
        voff_b[ps] = (n < d.N) ? (unsigned)((n * d.ldw + lslot * 8) * 2) : OOB_OFFSET;
    }

    const int ctot = d.c0 + d.c1;
    int kt_end = (split + 1) * p.kt_per_split;
    if (kt_end > p.kt_total) kt_end = p.kt_total;
    const int kt0 = split * p.kt_per_split * KSUB;  // in units of this kernel's K tile
    const int nk = kt_end * KSUB - kt0;
    const bool ragged = (d.K % BK) != 0;

    // gather state of the current (tap, source) segment: per-pass byte offset of the pixel row, or OOB
    unsigned voff_a[A_PASSES];
    int cur_seg = -1;
    auto load_tile = [&](int t, uint4* ra, uint4* rb) {
        const int kglob = t * BK;
        const int tap = kglob / ctot;
        int cc = kglob - tap * ctot;
        const bool second = cc >= d.c0;
        if (second) cc -= d.c0;
        const int seg = tap * 2 + (second ? 1 : 0);
        if (seg != cur_seg) {  // wave-uniform: new tap or switch to the concatenated source
            cur_seg = seg;
            const int ky = tap / d.ksize, kx = tap - ky * d.ksize;
            const int ld = second ? d.lda1 : d.lda0;
#pragma unroll
            for (int ps = 0; ps < A_PASSES; ++ps) {
                const int iy = a_iy0[ps] + ky, ix = a_ix0[ps] + kx;
                const bool ok = ((unsigned)iy < (unsigned)Hv) && ((unsigned)ix < (unsigned)Wv);
                const int pix = a_pix[ps] + (iy >> d.ups) * d.Win + (ix >> d.ups);
                voff_a[ps] = ok ? (unsigned)((pix * ld + lslot * 8) * 2) : OOB_OFFSET;
            }
        }
        const unsigned soff_a = (unsigned)cc * 2u, soff_b = (unsigned)kglob * 2u;
        if (ragged) {  // wave-uniform: only plain matrices whose K is not a multiple of 64
            const bool kbad = kglob + lslot * 8 >= d.K;
#pragma unroll
            for (int ps = 0; ps < A_PASSES; ++ps) ra[ps] = buf_load16(rs_a0, kbad ? OOB_OFFSET : voff_a[ps], soff_a);
#pragma unroll
            for (int ps = 0; ps < B_PASSES; ++ps) rb[ps] = buf_load16(rs_w, kbad ? OOB_OFFSET : voff_b[ps], soff_b);
        } else {
#pragma unroll
            for (int ps = 0; ps < A_PASSES; ++ps)
                ra[ps] = second ? buf_load16(rs_a1, voff_a[ps], soff_a) : buf_load16(rs_a0, voff_a[ps], soff_a);
#pragma unroll
            for (int ps = 0; ps < B_PASSES; ++ps) rb[ps] = buf_load16(rs_w, voff_b[ps], soff_b);
        }
    };
    // LDS addressing: (row >> 1) & 7 is the same for every 32-row pass / fragment of a lane, so each lane needs one
    // store offset and one read offset per k-step; passes and fragments are immediate offsets (32 rows = 4096 B).
    const int st_off = lds_off(lrow, lslot);
    auto store_tile = [&](int buf, const uint4* ra, const uint4* rb) {
        char* sa = smem + buf * STAGE_BYTES + st_off;
        char* sb = sa + BM * ROW_BYTES;
#pragma unroll
        for (int ps = 0; ps < A_PASSES; ++ps) *reinterpret_cast<uint4*>(sa + ps * RPP * ROW_BYTES) = ra[ps];
#pragma unroll
        for (int ps = 0; ps < B_PASSES; ++ps) *reinterpret_cast<uint4*>(sb + ps * RPP * ROW_BYTES) = rb[ps];
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

    int rd_a[KB / 16], rd_b[KB / 16];
#pragma unroll
    for (int ks = 0; ks < KB / 16; ++ks) {
        rd_a[ks] = lds_off_kb<KB>(wm * WM + l31, ks * 2 + hi);
        rd_b[ks] = BM * KROW_BYTES + lds_off_kb<KB>(wn * WN + l31, ks * 2 + hi);
    }
    auto compute_tile = [&](int buf) {
        const char* st = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < KB / 16; ++ks) {
            f16x8 af[MI], bf[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                U4H8 t;
#ifdef VD_GEMM_X_LDSBCAST  // experiment: every lane reads the same address (LDS bandwidth removed, instruction count kept)
                t.u = *reinterpret_cast<const uint4*>(st + i * 32 * KROW_BYTES + ks * 32);
#else
                t.u = *reinterpret_cast<const uint4*>(st + rd_a[ks] + i * 32 * KROW_BYTES);
#endif
                af[i] = t.h;
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                U4H8 t;
#ifdef VD_GEMM_X_LDSBCAST
                t.u = *reinterpret_cast<const uint4*>(st + BM * KROW_BYTES + j * 32 * KROW_BYTES + ks * 32);
#else
                t.u = *reinterpret_cast<const uint4*>(st + rd_b[ks] + j * 32 * KROW_BYTES);
#endif
                bf[j] = t.h;
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
    };

    if constexpr (STAGES == 0) {
    // ---- main loop: two register sets -> global loads run two K tiles ahead of the MFMAs, LDS double buffer,
        // one barrier per K tile.
        uint4 ra0[A_PASSES], rb0[B_PASSES], ra1[A_PASSES], rb1[B_PASSES];
        if (nk > 0) {
            load_tile(kt0, ra0, rb0);
            if (nk > 1) load_tile(kt0 + 1, ra1, rb1);
            store_tile(0, ra0, rb0);
        }
        __syncthreads();
        // Steady state has NO conditionals around the loads: the compiler's s_waitcnt vmcnt(N) before each LDS store
        // then only waits for the OLDER register set and leaves the 8 newest loads in flight across the barrier.
        int i = 0;
        for (; i + 3 < nk; i += 2) {
            load_tile(kt0 + i + 2, ra0, rb0);
            compute_tile(0);
            store_tile(1, ra1, rb1);
            __syncthreads();
            load_tile(kt0 + i + 3, ra1, rb1);
            compute_tile(1);
            store_tile(0, ra0, rb0);
            __syncthreads();
        }
        // tail: 1..3 tiles left; LDS stage 0 holds tile i, register set 1 holds tile i+1 (if any)
        const int left = nk - i;
        if (left >= 1) {
            if (left >= 3) load_tile(kt0 + i + 2, ra0, rb0);
            compute_tile(0);
            if (left >= 2) {
                store_tile(1, ra1, rb1);
                __syncthreads();
                compute_tile(1);
                if (left >= 3) {
                    store_tile(0, ra0, rb0);
                    __syncthreads();
                    compute_tile(0);
                }
            }
        }
        __syncthreads();
    } else {
        // ---- main loop, LDS-DMA form: STAGES LDS buffers, tiles are DMA'd STAGES-1 ahead, one barrier per K tile
        constexpr int LPT = A_PASSES + B_PASSES;  // DMA instructions per thread and tile
        const i32x4 ws_a0 = make_rsrc_words(reinterpret_cast<const f16*>(d.a0) + (size_t)z * d.stride_a, p.a0_bytes);
        const i32x4 ws_a1 = make_rsrc_words(d.a1 ? reinterpret_cast<const f16*>(d.a1) + (size_t)z * d.stride_a : d.a0,
                                            d.a1 ? p.a1_bytes : 0u);
        const i32x4 ws_w = make_rsrc_words(reinterpret_cast<const f16*>(d.w) + (size_t)z * d.stride_w, p.w_bytes);
        const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
        const unsigned wave_dst = (unsigned)(__builtin_amdgcn_readfirstlane(wave) * 1024);  // one wave-wide DMA = 64 x 16 bytes of rows
        const int sw_slot = lslot ^ lds_swz<KB>(lrow);  // the lane fetches the logical slot that lives at its physical slot
        unsigned dvoff_b[B_PASSES];
#pragma unroll
        for (int ps = 0; ps < B_PASSES; ++ps) {
            const int n = n0 + lrow + RPP * ps;
            dvoff_b[ps] = (n < d.N) ? (unsigned)((n * d.ldw + sw_slot * 8) * 2) : OOB_OFFSET;
        }
        unsigned dvoff_a[A_PASSES];
        // (tap, channel offset) of the next tile to issue, advanced incrementally: no per-tile integer division
        int n_tap = (kt0 * KB) / ctot;
        int n_cc = kt0 * KB - n_tap * ctot;
        int n_ky = n_tap / d.ksize, n_kx = n_tap - n_ky * d.ksize;
        bool seg_dirty = true;
        bool n_second = false;
        auto issue_tile = [&](int t, int buf) {
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
            const unsigned soff_a = (unsigned)((second ? n_cc - d.c0 : n_cc) * 2);
            const unsigned soff_b = (unsigned)(kglob * 2);
            const unsigned dst_a = lds0 + (unsigned)(buf * STAGE_BYTES) + wave_dst;
            const unsigned dst_b = dst_a + BM * KROW_BYTES;
            const i32x4 ra_src = second ? ws_a1 : ws_a0;
            if (ragged) {
                const bool kbad = kglob + sw_slot * 8 >= d.K;
#pragma unroll
                for (int ps = 0; ps < A_PASSES; ++ps)
                    dma16(ra_src, dst_a + ps * RPP * KROW_BYTES, kbad ? OOB_OFFSET : dvoff_a[ps], soff_a);
#pragma unroll
                for (int ps = 0; ps < B_PASSES; ++ps)
                    dma16(ws_w, dst_b + ps * RPP * KROW_BYTES, kbad ? OOB_OFFSET : dvoff_b[ps], soff_b);
            } else {
#pragma unroll
                for (int ps = 0; ps < A_PASSES; ++ps) dma16(ra_src, dst_a + ps * RPP * KROW_BYTES, dvoff_a[ps], soff_a);
#pragma unroll
                for (int ps = 0; ps < B_PASSES; ++ps) dma16(ws_w, dst_b + ps * RPP * KROW_BYTES, dvoff_b[ps], soff_b);
            }
            // advance to the next K tile
            n_cc += KB;
            if (n_cc >= ctot) {
                n_cc -= ctot;
                ++n_tap;
                ++n_kx;
                if (n_kx == d.ksize) { n_kx = 0; ++n_ky; }
                seg_dirty = true;
            }
        };
        constexpr int D = STAGES - 1;
#pragma unroll
        for (int j = 0; j < D; ++j)
            if (j < nk) issue_tile(kt0 + j, j);
        int cbuf = 0, ibuf = D % STAGES;
        for (int i = 0; i < nk; ++i) {
            // tile i must have landed; in steady state the D-1 younger tiles stay in flight across the barrier
            if (i + D - 1 < nk) wait_vmcnt<LPT * (D - 1)>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");  // LDS reads below must not be hoisted above the barrier
#ifdef VD_GEMM_X_NODMA  // experiment: no global traffic after the prologue (results are garbage)
            if (i + D < nk && i < 2) issue_tile(kt0 + i + D, ibuf);
#else
            if (i + D < nk) issue_tile(kt0 + i + D, ibuf);
#endif
            compute_tile(cbuf);
            cbuf = (cbuf + 1 == STAGES) ? 0 : cbuf + 1;
            ibuf = (ibuf + 1 == STAGES) ? 0 : ibuf + 1;
        }
        __syncthreads();
    }

    const EpiCtx e = make_epi(d, z);

    // ---- split-K / fp32 output: straight from registers (4 consecutive floats per lane and group)
    if (gridDim.y > 1 || (d.flags & VD_EPI_OUT_F32)) {
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
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                if (col + q < d.N) o[q] = v[q];
                        }
                    }
                }
        }
        return;
    }

    // ---- fused epilogue, part 1 (registers): + bias -> act / GEGLU -> * alpha -> fp16 into an LDS tile [BM][OUT_N]
    f16* cs = reinterpret_cast<f16*>(smem);
    const bool geglu = (d.act == VD_ACT_GEGLU);
    const int out_n0 = geglu ? tn * (BN / 2) : n0;

    // residual / row-vector segments of part 2 are requested NOW so their latency overlaps part 1 (the block is
    // short-lived on the K = 320..1280 projections: every serial memory round trip shows)
    constexpr int MAX_CH = BM * (BN / 8) / NT;
    uint4 pre_res[MAX_CH], pre_rv[MAX_CH];
    if (geglu) epi_prefetch<BM, BN / 16, NT, MAX_CH>(e, d.M, m0, out_n0, tid, pre_res, pre_rv);
    else epi_prefetch<BM, BN / 8, NT, MAX_CH>(e, d.M, m0, out_n0, tid, pre_res, pre_rv);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int lrow_c = wm * WM + i * 32 + l31;
        const int row = m0 + lrow_c;
        float bm = 0.f;
        if ((e.flags & VD_EPI_BIAS) && (e.flags & VD_EPI_BIAS_ALONG_M) && row < d.M) bm = (float)e.bias[row];
        if (geglu) {
            if constexpr (NI == 2) {
                // weight rows are packed per 64-row group as [32 value rows | 32 gate rows]: j = 0 value, j = 1 gate
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int lc = wn * (WN / 2) + 8 * g + 4 * hi;     // column inside the block's output tile
                    const int pn = n0 + wn * WN + 8 * g + 4 * hi;      // packed weight row of the value element
                    U2H4 bv, bg, o;
                    bv.u = make_uint2(0, 0);
                    bg.u = make_uint2(0, 0);
                    if (e.flags & VD_EPI_BIAS) {
                        bv.u = *reinterpret_cast<const uint2*>(e.bias + pn);
                        bg.u = *reinterpret_cast<const uint2*>(e.bias + pn + 32);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float v = acc[i][0][g * 4 + q] + (float)bv.e[q];
                        const float gt = acc[i][1][g * 4 + q] + (float)bg.e[q];
                        o.e[q] = (f16)(v * vd_gelu_erf(gt) * e.alpha);
                    }
                    *reinterpret_cast<uint2*>(cs + lrow_c * CS_LD + lc) = o.u;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int lc = wn * WN + j * 32 + 8 * g + 4 * hi;
                    const int col = n0 + lc;
                    float bq[4] = {bm, bm, bm, bm};
                    if ((e.flags & VD_EPI_BIAS) && !(e.flags & VD_EPI_BIAS_ALONG_M)) {
                        if (col + 4 <= d.N && (d.N & 3) == 0) {
                            U2H4 t;
                            t.u = *reinterpret_cast<const uint2*>(e.bias + col);
#pragma unroll
                            for (int q = 0; q < 4; ++q) bq[q] = (float)t.e[q];
                        } else {
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                if (col + q < d.N) bq[q] = (float)e.bias[col + q];
                        }
                    }
                    U2H4 o;
                    if (e.act == VD_ACT_NONE) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) o.e[q] = (f16)((acc[i][j][g * 4 + q] + bq[q]) * e.alpha);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) o.e[q] = (f16)(apply_act(e.act, acc[i][j][g * 4 + q] + bq[q]) * e.alpha);
                    }
                    *reinterpret_cast<uint2*>(cs + lrow_c * CS_LD + lc) = o.u;
                }
        }
    }
    __syncthreads();

    // ---- part 2: coalesced 16-byte row segments: (+ rowvec) (+ residual) -> global
    if (geglu) epi_writeout<BM, BN / 16, NT, MAX_CH, CS_LD>(e, d.M, m0, out_n0, tid, cs, pre_res, pre_rv);
    else epi_writeout<BM, BN / 8, NT, MAX_CH, CS_LD>(e, d.M, m0, out_n0, tid, cs, pre_res, pre_rv);
}

// Sum the split-K slabs and run the fused epilogue. One thread per 8 output columns.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs p, int nsplit) {
    const VdGemmDesc& d = p.d;
    const int z = blockIdx.z;
    const int chunks = (d.N + 7) / 8;
    const size_t total = (size_t)d.M * chunks;
    const EpiCtx e = make_epi(d, z);
    for (size_t c = (size_t)blockIdx.x * 256 + threadIdx.x; c < total; c += (size_t)gridDim.x * 256) {
        const int row = (int)(c / chunks);
        const int col = (int)(c - (size_t)row * chunks) * 8;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.f;
        const size_t slab = (size_t)d.M * d.N;
        const float* w0 = d.ws + ((size_t)z * nsplit * d.M + row) * (size_t)d.N + col;
        if (col + 8 <= d.N && (d.N & 3) == 0) {
            int s = 0;
            for (; s + 4 <= nsplit; s += 4) {  // 8 independent 16-byte loads in flight before the adds
                float4 x[4], y[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float* w = w0 + (size_t)(s + u) * slab;
                    x[u] = *reinterpret_cast<const float4*>(w);
                    y[u] = *reinterpret_cast<const float4*>(w + 4);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    v[0] += x[u].x; v[1] += x[u].y; v[2] += x[u].z; v[3] += x[u].w;
                    v[4] += y[u].x; v[5] += y[u].y; v[6] += y[u].z; v[7] += y[u].w;
                }
            }
            for (; s < nsplit; ++s) {
                const float* w = w0 + (size_t)s * slab;
                const float4 x = *reinterpret_cast<const float4*>(w);
                const float4 y = *reinterpret_cast<const float4*>(w + 4);
                v[0] += x.x; v[1] += x.y; v[2] += x.z; v[3] += x.w;
                v[4] += y.x; v[5] += y.y; v[6] += y.z; v[7] += y.w;
            }
        } else {
            for (int s = 0; s < nsplit; ++s)
                for (int i = 0; i < 8; ++i)
                    if (col + i < d.N) v[i] += w0[(size_t)s * slab + i];
        }
        epi_store8(e, row, col, v);
    }
}

template <int BM, int BN, int WM, int WN, int NT, int STAGES, int KB = 64>
int launch_cfg(const GemmArgs& a, int nsplit, hipStream_t stream) {
    constexpr int STAGE_BYTES = (BM + BN) * KB * 2;
    constexpr int EPI_BYTES = BM * (BN + 8) * 2;
    constexpr int NST = STAGES == 0 ? 2 : STAGES;
    constexpr int LDS = (NST * STAGE_BYTES > EPI_BYTES) ? NST * STAGE_BYTES : EPI_BYTES;
    static bool attr_done = false;
    if (!attr_done) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16_kernel<BM, BN, WM, WN, NT, STAGES, KB>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) {
            vd_set_error("vd_gemm_f16: cannot reserve %d bytes of LDS: %s", LDS, hipGetErrorString(e));
            return VD_ERR_LAUNCH;
        }
        attr_done = true;
    }
    dim3 grid(a.tiles_m * a.tiles_n, nsplit, a.d.batch > 0 ? a.d.batch : 1);
    hipLaunchKernelGGL((gemm_f16_kernel<BM, BN, WM, WN, NT, STAGES, KB>), grid, dim3(NT), LDS, stream, a);
    return vd_check_launch("vd_gemm_f16");
}

}  // namespace

extern "C" size_t vd_gemm_workspace_bytes(const VdGemmDesc* d) {
    // upper bound for any split factor the heuristic may choose
    const size_t batch = d->batch > 0 ? d->batch : 1;
    return batch * VD_MAX_SPLIT_K * (size_t)d->M * (size_t)d->N * sizeof(float);
}

namespace {
enum TileCfg { T128x128 = 0, T128x64 = 1, T64x64 = 2, T128x128w8 = 3, T128x64w8 = 4, T256x128 = 5, T128x256 = 6, T128x320 = 7 };

// validate + normalise the descriptor and pick tile shape / split factor
int plan_gemm(const VdGemmDesc* dp, GemmArgs& a, int& cfg_out, int& nsplit_out) {
    VD_REQUIRE(dp != nullptr, "vd_gemm_f16: null descriptor");
    a.d = *dp;
    VdGemmDesc& d = a.d;
    VD_REQUIRE(d.M > 0 && d.N > 0 && d.K > 0, "vd_gemm_f16: empty problem M=%d N=%d K=%d", d.M, d.N, d.K);
    VD_REQUIRE(d.K % 8 == 0, "vd_gemm_f16: K=%d must be a multiple of 8 (pad small-K operands with vd_im2col_small)", d.K);
    if (d.K % BK != 0)  // ragged K tail is zero-filled by the loader; only plain single-source matrices
        VD_REQUIRE(d.ksize <= 1 && d.a1 == nullptr, "vd_gemm_f16: conv/concat operands need K %% %d == 0 (K=%d)", BK, d.K);
    if (d.ksize <= 0) d.ksize = 1;
    if (d.stride <= 0) d.stride = 1;
    if (d.c0 <= 0) d.c0 = d.K / (d.ksize * d.ksize);
    if (d.a1 == nullptr) d.c1 = 0;
    const int ctot = d.c0 + d.c1;
    VD_REQUIRE(ctot * d.ksize * d.ksize == d.K, "vd_gemm_f16: K=%d != ksize^2*(c0+c1)=%d", d.K, ctot * d.ksize * d.ksize);
    VD_REQUIRE(d.c0 % 8 == 0 && d.c1 % 8 == 0, "vd_gemm_f16: channel counts must be multiples of 8");
    if (d.ksize > 1 || d.c1 > 0)
        VD_REQUIRE(d.c0 % BK == 0 && d.c1 % BK == 0, "vd_gemm_f16: conv/concat sources need channels %% %d == 0 (c0=%d c1=%d)", BK, d.c0, d.c1);
    if (d.lda0 <= 0) d.lda0 = d.c0;
    if (d.lda1 <= 0) d.lda1 = d.c1;
    if (d.ldw <= 0) d.ldw = d.K;
    if (d.Hout <= 0 || d.Wout <= 0) {  // plain matrix: one "pixel" per row
        d.Hin = 1; d.Win = d.M; d.Hout = 1; d.Wout = d.M; d.pad = 0; d.stride = 1; d.ups = 0;
    }
    VD_REQUIRE(d.M % (d.Hout * d.Wout) == 0, "vd_gemm_f16: M=%d is not a multiple of Hout*Wout=%d", d.M, d.Hout * d.Wout);
    const int n_out = (d.act == VD_ACT_GEGLU) ? d.N / 2 : d.N;
    if (d.ldc <= 0) d.ldc = n_out;
    if (d.ldr <= 0) d.ldr = n_out;
    if (d.batch <= 0) d.batch = 1;
    if (d.flags & VD_EPI_BIAS) VD_REQUIRE(d.bias != nullptr, "vd_gemm_f16: bias flag without pointer");
    if (d.flags & VD_EPI_ROWVEC) VD_REQUIRE(d.rowvec != nullptr && d.rows_per_batch > 0, "vd_gemm_f16: rowvec flag without pointer/rows_per_batch");
    if (d.flags & VD_EPI_RESIDUAL) VD_REQUIRE(d.res != nullptr, "vd_gemm_f16: residual flag without pointer");
    VD_REQUIRE(d.a0 && d.w && d.out, "vd_gemm_f16: null operand");
    if (d.act == VD_ACT_GEGLU) VD_REQUIRE(d.N % 128 == 0, "vd_gemm_f16: GEGLU needs N %% 128 == 0");
    if (d.flags & VD_EPI_ROWVEC)
        VD_REQUIRE(d.act == VD_ACT_NONE && d.alpha == 1.0f, "vd_gemm_f16: rowvec epilogue requires act=none, alpha=1");
    if (d.flags & VD_EPI_OUT_F32)
        VD_REQUIRE(!(d.flags & (VD_EPI_ROWVEC | VD_EPI_RESIDUAL)) && d.act != VD_ACT_GEGLU, "vd_gemm_f16: fp32 output supports bias/act/alpha only");

    a.kt_total = (d.K + BK - 1) / BK;
    {
        const size_t in_rows = (size_t)(d.M / (d.Hout * d.Wout)) * d.Hin * d.Win;
        const size_t a0b = in_rows * (size_t)d.lda0 * 2, a1b = d.a1 ? in_rows * (size_t)d.lda1 * 2 : 0;
        const size_t wb = (size_t)d.N * d.ldw * 2;
        VD_REQUIRE(a0b < (1ull << 31) && a1b < (1ull << 31) && wb < (1ull << 31),
                   "vd_gemm_f16: operand larger than 2 GiB per batch entry (32-bit buffer offsets); split the batch");
        a.a0_bytes = (unsigned)a0b;
        a.a1_bytes = (unsigned)a1b;
        a.w_bytes = (unsigned)wb;
        a.plain = (d.ksize == 1 && d.stride == 1 && d.pad == 0 && d.ups == 0 && d.Hin * d.Win == d.Hout * d.Wout) ? 1 : 0;
    }

    // ---- tile / split choice: a small cost model, calibrated on MI355X with tools/gemm_sweep.py.
    // A launch runs in "rounds" of `cap` co-resident blocks (LDS / VGPR limits per CU x 256 CUs); the time of one
    // 64-deep K tile of one block depends on how full the CUs are (t_solo: <= 1 block per CU .. t_full: every slot
    // taken; the chip also clocks higher when part of it idles).  Split-K adds the fp32 slab round trip + one reduce
    // launch.  What this buys over fixed thresholds: grids that just overflow a round (e.g. 640 blocks on 512 slots
    // ran 2 rounds at 25 % more time than 480 blocks in one) are avoided.
    struct Cand { TileCfg cfg; int bm, bn, cap; float t_solo, t_full, lo, hi, fix; };
    // 128x320 (8 waves of 32x160, one block per CU): the N = 320 layers of the 64x64 level read the activation panel
    // once instead of 5x (94 FLOP per byte fetched vs 43 for 128x64) and M = 32768 gives exactly 256 blocks -- one
    // round, no tail: 90 -> 75 us on the 3x3 convs there.  No split-K, no GEGLU (odd number of 32-column blocks per wave).
    static const Cand cands[4] = {
        {T128x128, 128, 128, 512, 0.75f, 1.10f, 0.50f, 1.00f, 6.5f},
        {T128x64, 128, 64, 768, 0.50f, 1.00f, 0.25f, 0.83f, 4.f},
        {T64x64, 64, 64, 1024, 0.38f, 0.78f, 0.25f, 0.75f, 3.f},
        {T128x320, 128, 320, 256, 1.20f, 1.50f, 0.50f, 1.00f, 6.f}};
    const int zb = d.batch;
    const bool can_split = (d.ws != nullptr || d.split_k > 1) && d.act != VD_ACT_GEGLU && !(d.flags & VD_EPI_OUT_F32);
    auto model_us = [&](const Cand& c, int ns) {
        const int tiles = ((d.M + c.bm - 1) / c.bm) * ((d.N + c.bn - 1) / c.bn) * zb;
        const long blocks = (long)tiles * ns;
        const int kt = (a.kt_total + ns - 1) / ns;
        const long full = blocks / c.cap, rem = blocks % c.cap;
        float t = (float)full * (kt * c.t_full + c.fix);
        if (rem) {
            const float load = (float)rem / c.cap;
            const float f = load <= c.lo ? 0.f : (load >= c.hi ? 1.f : (load - c.lo) / (c.hi - c.lo));
            t += kt * (c.t_solo + f * (c.t_full - c.t_solo)) + c.fix;
        }
        if (ns > 1) t += 3.f + 2.f * ns * (float)d.M * d.N * zb * 4.f / 8.0e6f + 2.f;  // slabs at ~8 TB/s (cache resident) + launch
        return t;
    };
    TileCfg cfg = T64x64;
    int nsplit = 1;
    if (d.act == VD_ACT_GEGLU) cfg = T128x128w8;  // 8 waves: the erf-heavy epilogue of one wave overlaps MFMAs of others
    else if (d.M < 96 || d.N < 96) {
        // small-M weight streaming (time-embedding MLPs, the 0-D text-latent flow: M = CFG batch, N x K up to 5120 x
        // 10240): 64x64 tiles, but split K until the grid covers the chip -- the weight matrix is the only traffic
        cfg = T64x64;
        float best = 1e30f;
        const int ns_max = (d.split_k > 0) ? d.split_k : ((can_split && a.kt_total >= 16) ? VD_MAX_SPLIT_K / 2 : 1);
        for (int ns = (d.split_k > 0 ? d.split_k : 1); ns <= ns_max; ++ns) {
            if (ns > 1 && a.kt_total / ns < 4) break;
            const float t = model_us(cands[2], ns);
            if (t < best) { best = t; nsplit = ns; }
        }
    } else {
        float best = 1e30f;
        const int ns_max = (d.split_k > 0) ? d.split_k : ((can_split && a.kt_total >= 32) ? VD_MAX_SPLIT_K / 2 : 1);
        for (const Cand& c : cands)
            for (int ns = (d.split_k > 0 ? d.split_k : 1); ns <= ns_max; ++ns) {
                if (ns > 1 && (a.kt_total / ns < 8 || c.cfg == T128x320)) break;
                const float t = model_us(c, ns);
                if (t < best) { best = t; cfg = c.cfg; nsplit = ns; }
            }
        // (256x128 / 128x256 block tiles -- 25 % fewer bytes L2 -> LDS per FLOP, 8 waves, one block per CU -- reach 995 vs
        //  895 TF/s at 8192^3 but measure equal on the VAE decoder's large convs (N = 128..512: the activation panel is
        //  read by 1-4 column tiles only) and equal or slower on every UNet shape: kept for VD_GEMM_TILE=5|6 experiments,
        //  never chosen here.)
    }
    {   // developer override for tile experiments: VD_GEMM_TILE=0|1|2 (never set in production runs)
        static const char* ov = getenv("VD_GEMM_TILE");
        if (ov && (d.act != VD_ACT_GEGLU || ov[0] == '0' || ov[0] == '3' || ov[0] == '5' || ov[0] == '6')) {
            cfg = (TileCfg)(ov[0] - '0');
            if (cfg == T128x320) {
                nsplit = 1;
            } else if (cfg == T256x128 || cfg == T128x256) {
                nsplit = 1;
                if (d.split_k <= 0 && can_split && a.kt_total >= 32) {
                    const int bm2 = cfg == T256x128 ? 256 : 128, bn2 = cfg == T256x128 ? 128 : 256;
                    const int tiles = ((d.M + bm2 - 1) / bm2) * ((d.N + bn2 - 1) / bn2) * zb;
                    nsplit = 256 / tiles;  // one block per CU
                    if (nsplit < 1) nsplit = 1;
                    while (nsplit > 1 && a.kt_total / nsplit < 8) --nsplit;
                }
            } else
            if (d.split_k <= 0 && cfg <= T64x64) {  // re-plan the split for the forced tile
                float best = 1e30f;
                const int ns_max = (can_split && a.kt_total >= 32) ? VD_MAX_SPLIT_K / 2 : 1;
                for (int ns = 1; ns <= ns_max; ++ns) {
                    if (ns > 1 && a.kt_total / ns < 8) break;
                    const float t = model_us(cands[(int)cfg], ns);
                    if (t < best) { best = t; nsplit = ns; }
                }
            }
        }
    }
    if (d.split_k > 0) nsplit = d.split_k;
    int bm = 128, bn = 128;
    if (cfg == T128x64 || cfg == T128x64w8) { bm = 128; bn = 64; }
    if (cfg == T64x64) { bm = 64; bn = 64; }
    if (cfg == T256x128) { bm = 256; bn = 128; }
    if (cfg == T128x256) { bm = 128; bn = 256; }
    if (cfg == T128x320) { bm = 128; bn = 320; }
    a.tiles_m = (d.M + bm - 1) / bm;
    a.tiles_n = (d.N + bn - 1) / bn;
    if (nsplit > a.kt_total) nsplit = a.kt_total;
    if (nsplit > 1) {
        VD_REQUIRE(d.ws != nullptr, "vd_gemm_f16: split_k=%d needs a workspace", nsplit);
        VD_REQUIRE(d.act != VD_ACT_GEGLU, "vd_gemm_f16: split-K with GEGLU epilogue unsupported");
        VD_REQUIRE(!(d.flags & VD_EPI_OUT_F32) || true, "unreachable");
        VD_REQUIRE(nsplit <= VD_MAX_SPLIT_K, "vd_gemm_f16: split_k=%d > %d", nsplit, VD_MAX_SPLIT_K);
    }
    a.kt_per_split = (a.kt_total + nsplit - 1) / nsplit;
    nsplit = (a.kt_total + a.kt_per_split - 1) / a.kt_per_split;
    cfg_out = (int)cfg;
    nsplit_out = nsplit;
    return VD_OK;
}
}  // namespace

extern "C" int vd_gemm_plan(const VdGemmDesc* dp, int* tile_cfg, int* nsplit) {
    GemmArgs a;
    int c = 0, n = 1;
    const int rc = plan_gemm(dp, a, c, n);
    if (rc != VD_OK) return rc;
    if (tile_cfg) *tile_cfg = c;
    if (nsplit) *nsplit = n;
    return VD_OK;
}

extern "C" int vd_gemm_f16(const VdGemmDesc* dp, hipStream_t stream) {
    GemmArgs a;
    int cfg = 0, nsplit = 1;
    int rc = plan_gemm(dp, a, cfg, nsplit);
    if (rc != VD_OK) return rc;
    const VdGemmDesc& d = a.d;
    const int zb = d.batch;

    static const char* dma_env = getenv("VD_GEMM_DMA");  // developer switch: 0 = register staging, 2 = LDS-DMA (3 stages measured slower: occupancy)
    const int dma = dma_env ? (dma_env[0] - '0') : VD_GEMM_DEFAULT_DMA;
#define VD_LAUNCH(BM_, BN_, WM_, WN_, NT_)                                                        \
    (dma == 2 ? launch_cfg<BM_, BN_, WM_, WN_, NT_, 2>(a, nsplit, stream)                         \
              : launch_cfg<BM_, BN_, WM_, WN_, NT_, 0>(a, nsplit, stream))
    // grids that cannot fill the CUs twice over are latency-bound per block: give those a deeper DMA ring instead
    const int grid_blocks = a.tiles_m * a.tiles_n * nsplit * zb;
    static const char* deep_env = getenv("VD_GEMM_DEEP");
    const int deep = deep_env ? (deep_env[0] - '0') : VD_GEMM_DEFAULT_DEEP;
    static const char* deep_blk_env = getenv("VD_GEMM_DEEP_MAXBLK");
    const int deep_maxblk = deep_blk_env ? atoi(deep_blk_env) : 400;
    // (32-deep K tiles with a 4- or 5-stage ring -- launch_cfg<..., STAGES, 32> -- were measured 15-20 % slower than
    // 64-deep / 2 stages on every UNet shape and on 4096^3 / 8192^3: the extra barrier per 32-deep step costs more than
    // the deeper prefetch buys.  The kernel keeps the KB parameter; no 32-deep instance is built.)
    if (dma == 2 && deep >= 3 && grid_blocks <= deep_maxblk && a.kt_per_split >= (deep_blk_env ? 3 : 8) && (cfg == T128x64 || cfg == T64x64)) {
        if (cfg == T128x64) rc = deep == 4 ? launch_cfg<128, 64, 64, 32, 256, 4>(a, nsplit, stream) : launch_cfg<128, 64, 64, 32, 256, 3>(a, nsplit, stream);
        else rc = deep == 4 ? launch_cfg<64, 64, 32, 32, 256, 4>(a, nsplit, stream) : launch_cfg<64, 64, 32, 32, 256, 3>(a, nsplit, stream);
    } else
    switch (cfg) {
        case T128x128: rc = VD_LAUNCH(128, 128, 64, 64, 256); break;
        case T128x64: rc = VD_LAUNCH(128, 64, 64, 32, 256); break;
        case T128x128w8: rc = VD_LAUNCH(128, 128, 32, 64, 512); break;
        case T128x64w8: rc = VD_LAUNCH(128, 64, 32, 32, 512); break;
        case T256x128: rc = launch_cfg<256, 128, 64, 64, 512, 2>(a, nsplit, stream); break;
        case T128x256: rc = launch_cfg<128, 256, 64, 64, 512, 2>(a, nsplit, stream); break;
        case T128x320: rc = launch_cfg<128, 320, 32, 160, 512, 2>(a, nsplit, stream); break;
        default: rc = VD_LAUNCH(64, 64, 32, 32, 256); break;
    }
#undef VD_LAUNCH
    if (rc != VD_OK) return rc;
    if (nsplit > 1) {
        const size_t total = (size_t)d.M * ((d.N + 7) / 8);
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks, 1, zb), dim3(256), 0, stream, a, nsplit);
        return vd_check_launch("vd_gemm_f16/splitk_reduce");
    }
    return VD_OK;
}
