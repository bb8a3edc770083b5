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

namespace {

constexpr int BK = 64;           // K tile (halfs); one LDS row = 128 bytes = 8 x 16-byte slots
constexpr int ROW_BYTES = BK * 2;

struct GemmArgs {
    VdGemmDesc d;
    int tiles_m, tiles_n, kt_total, kt_per_split;
};

// swizzled byte offset of (row r, 16-byte slot s) inside a [rows][64] f16 LDS tile
__device__ __forceinline__ int lds_off(int r, int s) { return r * ROW_BYTES + ((s ^ ((r >> 1) & 7)) << 4); }

struct EpiCtx {
    const f16* bias;
    const f16* rowvec;
    const f16* res;
    void* out;
    int N, ldc, ldr, rows_per_batch, flags, act;
    float alpha;
};

// Apply the fused epilogue to 8 consecutive output columns of one row and store them.
// Order: v = acc (+bias) (+rowvec[batch]) -> act -> *alpha -> (+residual).
__device__ __forceinline__ void epi_store8(const EpiCtx& e, int row, int col, float* v, const float* g) {
    const bool full = (col + 8 <= e.N) && ((e.N & 7) == 0);
    float b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = 0.f;
    if (e.flags & VD_EPI_BIAS) {
        if (e.flags & VD_EPI_BIAS_ALONG_M) {
            const float bv = (float)e.bias[row];
#pragma unroll
            for (int i = 0; i < 8; ++i) b[i] = bv;
        } else if (full) {
            U4H8 t;
            t.u = *reinterpret_cast<const uint4*>(e.bias + col);
#pragma unroll
            for (int i = 0; i < 8; ++i) b[i] = (float)t.e[i];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (col + i < e.N) b[i] = (float)e.bias[col + i];
        }
    }
    if (e.flags & VD_EPI_ROWVEC) {
        const f16* rv = e.rowvec + (size_t)(row / e.rows_per_batch) * e.N + col;
        if (full) {
            U4H8 t;
            t.u = *reinterpret_cast<const uint4*>(rv);
#pragma unroll
            for (int i = 0; i < 8; ++i) b[i] += (float)t.e[i];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (col + i < e.N) b[i] += (float)rv[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += b[i];
    if (e.act == VD_ACT_GEGLU) {
        // g = gate pre-activations (bias for the gate half is folded by the caller)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = v[i] * vd_gelu_erf(g[i]);
    } else if (e.act == VD_ACT_QUICK_GELU) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = vd_quick_gelu(v[i]);
    } else if (e.act == VD_ACT_SILU) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = vd_silu(v[i]);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= e.alpha;
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
    if (e.flags & VD_EPI_OUT_F32) {
        float* op = reinterpret_cast<float*>(e.out) + (size_t)row * e.ldc + col;
        if (full && ((e.ldc & 3) == 0)) {
            *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (col + i < e.N) op[i] = v[i];
        }
    } else {
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
}

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

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_f16_kernel(const GemmArgs p) {
    constexpr int WAVES_N = BN / WN;
    constexpr int WAVES_M = BM / WM;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per block");
    constexpr int MI = WM / 32, NI = WN / 32;
    constexpr int A_PASSES = BM / 32, B_PASSES = BN / 32;
    constexpr int STAGE_BYTES = (BM + BN) * ROW_BYTES;
    constexpr int CS_LD = BN + 4;  // fp32 epilogue tile leading dimension

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

    const f16* a0 = reinterpret_cast<const f16*>(d.a0) + (size_t)z * d.stride_a;
    const f16* a1 = reinterpret_cast<const f16*>(d.a1) + (size_t)z * d.stride_a;
    const f16* wp = reinterpret_cast<const f16*>(d.w) + (size_t)z * d.stride_w;

    // ---- per-thread gather coordinates: 8 threads per 128-byte row, 32 rows per pass
    const int lrow = tid >> 3, lslot = tid & 7;
    int a_iy0[A_PASSES], a_ix0[A_PASSES], a_pix[A_PASSES];
    const int HWo = d.Hout * d.Wout;
    const int Hv = d.Hin << d.ups, Wv = d.Win << d.ups;
#pragma unroll
    for (int ps = 0; ps < A_PASSES; ++ps) {
        const int m = m0 + lrow + 32 * ps;
        if (m < d.M) {
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
    int kt = split * p.kt_per_split;
    int kt_end = kt + p.kt_per_split;
    if (kt_end > p.kt_total) kt_end = p.kt_total;

    uint4 ra[A_PASSES], rb[B_PASSES];

    auto load_tile = [&](int t) {
        const int kglob = t * BK;
        const int tap = kglob / ctot;
        int cc = kglob - tap * ctot;
        const int ky = tap / d.ksize, kx = tap - ky * d.ksize;
        const f16* src = a0;
        int ld = d.lda0;
        if (cc >= d.c0) {
            src = a1;
            ld = d.lda1;
            cc -= d.c0;
        }
#pragma unroll
        for (int ps = 0; ps < A_PASSES; ++ps) {
            const int iy = a_iy0[ps] + ky, ix = a_ix0[ps] + kx;
            const bool ok = ((unsigned)iy < (unsigned)Hv) && ((unsigned)ix < (unsigned)Wv) && (kglob + lslot * 8 < d.K);
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ok) {
                const size_t pix = (size_t)a_pix[ps] + (size_t)((iy >> d.ups) * d.Win + (ix >> d.ups));
                v = *reinterpret_cast<const uint4*>(src + pix * ld + cc + lslot * 8);
            }
            ra[ps] = v;
        }
#pragma unroll
        for (int ps = 0; ps < B_PASSES; ++ps) {
            const int n = n0 + lrow + 32 * ps;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (n < d.N && kglob + lslot * 8 < d.K)
                v = *reinterpret_cast<const uint4*>(wp + (size_t)n * d.ldw + kglob + lslot * 8);
            rb[ps] = v;
        }
    };
    auto store_tile = [&](int buf) {
        char* sa = smem + buf * STAGE_BYTES;
        char* sb = sa + BM * ROW_BYTES;
#pragma unroll
        for (int ps = 0; ps < A_PASSES; ++ps) *reinterpret_cast<uint4*>(sa + lds_off(lrow + 32 * ps, lslot)) = ra[ps];
#pragma unroll
        for (int ps = 0; ps < B_PASSES; ++ps) *reinterpret_cast<uint4*>(sb + lds_off(lrow + 32 * ps, lslot)) = rb[ps];
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (kt < kt_end) {
        load_tile(kt);
        store_tile(0);
    }
    __syncthreads();

    int buf = 0;
    for (; kt < kt_end; ++kt) {
        const bool more = (kt + 1) < kt_end;
        if (more) load_tile(kt + 1);  // global loads stay in flight under the MFMAs below

        const char* sa = smem + buf * STAGE_BYTES;
        const char* sb = sa + BM * ROW_BYTES;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            f16x8 af[MI], bf[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                U4H8 t;
                t.u = *reinterpret_cast<const uint4*>(sa + lds_off(wm * WM + i * 32 + l31, ks * 2 + hi));
                af[i] = t.h;
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                U4H8 t;
                t.u = *reinterpret_cast<const uint4*>(sb + lds_off(wn * WN + j * 32 + l31, ks * 2 + hi));
                bf[j] = t.h;
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    // ---- epilogue: accumulators -> LDS (fp32) -> coalesced 16-byte row segments
    float* cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const int col = wn * WN + j * 32 + l31;
                cs[row * CS_LD + col] = acc[i][j][r];
            }
    __syncthreads();

    if (gridDim.y > 1) {
        // split-K: raw fp32 partial slab [split][M][N]; the reduce kernel applies the epilogue
        float* ws = d.ws + ((size_t)z * gridDim.y + split) * (size_t)d.M * d.N;
        constexpr int CH = BN / 4;
        for (int c = tid; c < BM * CH; c += 256) {
            const int r = c / CH, cc = (c - r * CH) * 4;
            const int row = m0 + r, col = n0 + cc;
            if (row < d.M && col < d.N) {
                const float* s = cs + r * CS_LD + cc;
                float* o = ws + (size_t)row * d.N + col;
                if (col + 4 <= d.N && (d.N & 3) == 0) {
                    *reinterpret_cast<float4*>(o) = make_float4(s[0], s[1], s[2], s[3]);
                } else {
                    for (int i = 0; i < 4; ++i)
                        if (col + i < d.N) o[i] = s[i];
                }
            }
        }
        return;
    }

    const EpiCtx e = make_epi(d, z);
    if (d.act == VD_ACT_GEGLU) {
        // weight rows are packed per 128-row group as [64 value rows | 64 gate rows]
        constexpr int HALF = BN / 2;
        constexpr int CH = HALF / 8;
        for (int c = tid; c < BM * CH; c += 256) {
            const int r = c / CH, cc = (c - r * CH) * 8;
            const int row = m0 + r, col = tn * HALF + cc;
            if (row < d.M && col < e.N) {
                float v[8], g[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    v[i] = cs[r * CS_LD + cc + i];
                    g[i] = cs[r * CS_LD + HALF + cc + i];
                }
                if (d.flags & VD_EPI_BIAS) {
                    // packed bias layout mirrors the packed weight rows
                    const f16* bp = reinterpret_cast<const f16*>(d.bias) + n0;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        v[i] += (float)bp[cc + i];
                        g[i] += (float)bp[HALF + cc + i];
                    }
                }
                EpiCtx e2 = e;
                e2.flags &= ~VD_EPI_BIAS;
                epi_store8(e2, row, col, v, g);
            }
        }
    } else {
        constexpr int CH = BN / 8;
        for (int c = tid; c < BM * CH; c += 256) {
            const int r = c / CH, cc = (c - r * CH) * 8;
            const int row = m0 + r, col = n0 + cc;
            if (row < d.M && col < d.N) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = cs[r * CS_LD + cc + i];
                epi_store8(e, row, col, v, nullptr);
            }
        }
    }
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
        for (int s = 0; s < nsplit; ++s) {
            const float* w = d.ws + (((size_t)z * nsplit + s) * d.M + row) * (size_t)d.N + col;
            if (col + 8 <= d.N && (d.N & 3) == 0) {
                const float4 x = *reinterpret_cast<const float4*>(w);
                const float4 y = *reinterpret_cast<const float4*>(w + 4);
                v[0] += x.x; v[1] += x.y; v[2] += x.z; v[3] += x.w;
                v[4] += y.x; v[5] += y.y; v[6] += y.z; v[7] += y.w;
            } else {
                for (int i = 0; i < 8; ++i)
                    if (col + i < d.N) v[i] += w[i];
            }
        }
        epi_store8(e, row, col, v, nullptr);
    }
}

template <int BM, int BN, int WM, int WN>
int launch_cfg(const GemmArgs& a, int nsplit, hipStream_t stream) {
    constexpr int STAGE_BYTES = (BM + BN) * ROW_BYTES;
    constexpr int EPI_BYTES = BM * (BN + 4) * 4;
    constexpr int LDS = (2 * STAGE_BYTES > EPI_BYTES) ? 2 * STAGE_BYTES : EPI_BYTES;
    static bool attr_done = false;
    if (!attr_done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16_kernel<BM, BN, WM, WN>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_done = true;
    }
    dim3 grid(a.tiles_m * a.tiles_n, nsplit, a.d.batch > 0 ? a.d.batch : 1);
    hipLaunchKernelGGL((gemm_f16_kernel<BM, BN, WM, WN>), grid, dim3(256), LDS, stream, a);
    return vd_check_launch("vd_gemm_f16");
}

}  // namespace

extern "C" size_t vd_gemm_workspace_bytes(const VdGemmDesc* d) {
    // upper bound for any split factor the heuristic may choose (<= 16)
    const size_t batch = d->batch > 0 ? d->batch : 1;
    return batch * 16 * (size_t)d->M * (size_t)d->N * sizeof(float);
}

namespace {
enum TileCfg { T128x128 = 0, T128x64 = 1, T64x64 = 2 };

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

    a.kt_total = (d.K + BK - 1) / BK;

    // ---- tile / split heuristic: fill >= ~1 wave of the 256 CUs (2 blocks per CU resident)
    TileCfg cfg;
    auto tiles = [&](int bm, int bn) { return ((d.M + bm - 1) / bm) * ((d.N + bn - 1) / bn); };
    const int zb = d.batch;
    if (d.act == VD_ACT_GEGLU) cfg = T128x128;
    else if (d.N % 128 == 0 && tiles(128, 128) * zb >= 384) cfg = T128x128;
    else if (tiles(128, 64) * zb >= 256 && d.M >= 128) cfg = T128x64;
    else if (d.N % 128 == 0 && d.N >= 128 && d.M >= 128 && tiles(64, 64) * zb > 2048) cfg = T128x128;
    else cfg = T64x64;
    int bm = 128, bn = 128;
    if (cfg == T128x64) { bm = 128; bn = 64; }
    if (cfg == T64x64) { bm = 64; bn = 64; }
    a.tiles_m = (d.M + bm - 1) / bm;
    a.tiles_n = (d.N + bn - 1) / bn;

    int nsplit = 1;
    if (d.split_k > 0) nsplit = d.split_k;
    else if (d.ws != nullptr && d.act != VD_ACT_GEGLU) {
        const int nblk = a.tiles_m * a.tiles_n * zb;
        if (nblk < 192 && a.kt_total >= 16) {
            nsplit = (384 + nblk - 1) / nblk;
            if (nsplit > 16) nsplit = 16;
            while (nsplit > 1 && a.kt_total / nsplit < 8) --nsplit;
        }
    }
    if (nsplit > a.kt_total) nsplit = a.kt_total;
    if (nsplit > 1) {
        VD_REQUIRE(d.ws != nullptr, "vd_gemm_f16: split_k=%d needs a workspace", nsplit);
        VD_REQUIRE(d.act != VD_ACT_GEGLU, "vd_gemm_f16: split-K with GEGLU epilogue unsupported");
        VD_REQUIRE(nsplit <= 16, "vd_gemm_f16: split_k=%d > 16", nsplit);
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

    switch (cfg) {
        case T128x128: rc = launch_cfg<128, 128, 64, 64>(a, nsplit, stream); break;
        case T128x64: rc = launch_cfg<128, 64, 64, 32>(a, nsplit, stream); break;
        default: rc = launch_cfg<64, 64, 32, 32>(a, nsplit, stream); break;
    }
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
