// vd_gemm_f16: validation, launch planner, the two-waves-per-SIMD instances of the kernel template in
// gemm_kernel.h and the split-K reduce kernel.  (One-wave-per-SIMD instances: gemm_big.hip.)
#include "gemm_kernel.h"
#include "conv_halo_kernel.h"
#include <mutex>
#include <vector>

// conv_halo.hip
int vd_conv_halo_plan(const void* gemm_args, int can_split, void* conv_args, int* variant_out, int* nsplit_out);
int vd_conv_halo_launch(const void* conv_args, int variant, int nsplit, hipStream_t stream);
const char* vd_conv_halo_name(int variant);

namespace {

// Sum the split-K slabs and run the fused epilogue. One thread per 8 output columns.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmArgs p, int nsplit) {
    const VdGemmDesc& d = p.d;
    const int z = blockIdx.z;
    const int chunks = (d.N + 7) / 8;
    const size_t total = (size_t)d.M * chunks;
    const EpiCtx e = make_epi(d, z);
    const bool fast = epi8_fast(e);
    for (size_t c = (size_t)blockIdx.x * 256 + threadIdx.x; c < total; c += (size_t)gridDim.x * 256) {
        const int row = (int)(c / chunks);
        const int col = (int)(c - (size_t)row * chunks) * 8;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.f;
        const size_t slab = (size_t)d.M * d.N;
        const float* w0 = d.ws + ((size_t)z * nsplit * d.M + row) * (size_t)d.N + col;
        Epi8Ops ops;
        epi8_request(e, row, col, ops, fast);   // (N % 8 == 0: every group is whole) -- in flight across the slab reads
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
        if (fast) epi8_finish(e, row, col, v, ops);
        else epi_store8(e, row, col, v);
    }
}

// The same reduction for outputs that feed a GroupNorm (VdGemmDesc.out_stats): a block owns 64 rows x (8 OCT) columns, sums the
// slabs, runs the fused epilogue and emits per-channel (mean, M2) of the 64 values it stored per channel (csrc/gn_fused.hip).
// grid (ceil(N / (8 OCT)), M / 64), 256 threads = OCT column octets x 256 / OCT row lanes, RPT = 64 OCT / 256 rows per thread.
// OCT = 8 (OCT = 16 -- 512-byte row pieces, 16 loads in flight per slab pair -- measured slower).
template <int OCT>
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(const GemmArgs p, int nsplit) {
    constexpr int LANES = 256 / OCT, RPT = 64 / LANES, COLS = OCT * 8;
    // partial sums as [column-in-octet i][S | Q][row lane][octet] planes of LANES * OCT (+ 4: the planes start 4 banks apart) words: a
    // wave's 64 stores of one (i, S|Q) are 64 consecutive words, and the 64 column sums read planes that differ in (i, octet).
    // (Round 5's red[row lane][column][2] put a wave's stores of one i on TWO banks: 0.56 of this kernel's LDS cycles were conflicts.)
    constexpr int PLANE = LANES * OCT + 4;
    __shared__ float red[8 * 2 * PLANE];
    __shared__ float pivot[COLS];
    const VdGemmDesc& d = p.d;
    const EpiCtx e = make_epi(d, 0);
    const int tid = threadIdx.x, co = tid % OCT, rl = tid / OCT;
    const int col = blockIdx.x * COLS + co * 8;
    const int row0 = blockIdx.y * 64;
    const size_t slab = (size_t)d.M * d.N;
    float fin[RPT][8];
#pragma unroll
    for (int u = 0; u < RPT; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) fin[u][i] = 0.f;
    if (col < d.N) {   // N % 8 == 0: whole octets
        float v[RPT][8];
#pragma unroll
        for (int u = 0; u < RPT; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) v[u][i] = 0.f;
        const float* w0 = d.ws + (size_t)(row0 + rl) * d.N + col;
        const bool fast = epi8_fast(e);
        Epi8Ops ops[RPT];
        // bias / row vector / residual of the thread's RPT row pieces: in flight across the slab reads
#pragma unroll
        for (int u = 0; u < RPT; ++u) epi8_request(e, row0 + rl + LANES * u, col, ops[u], fast);
        int s = 0;
        for (; s + 4 <= nsplit; s += 4) {   // 8 RPT independent 16-byte loads in flight before the adds (slab order kept)
            float4 x[4][RPT], y[4][RPT];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int u = 0; u < RPT; ++u) {
                    const float* w = w0 + (size_t)(s + t) * slab + (size_t)(LANES * u) * d.N;
                    x[t][u] = *reinterpret_cast<const float4*>(w);
                    y[t][u] = *reinterpret_cast<const float4*>(w + 4);
                }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int u = 0; u < RPT; ++u) {
                    v[u][0] += x[t][u].x; v[u][1] += x[t][u].y; v[u][2] += x[t][u].z; v[u][3] += x[t][u].w;
                    v[u][4] += y[t][u].x; v[u][5] += y[t][u].y; v[u][6] += y[t][u].z; v[u][7] += y[t][u].w;
                }
        }
        for (; s + 2 <= nsplit; s += 2) {   // 4 RPT independent 16-byte loads in flight before the adds
            float4 x[2][RPT], y[2][RPT];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < RPT; ++u) {
                    const float* w = w0 + (size_t)(s + t) * slab + (size_t)(LANES * u) * d.N;
                    x[t][u] = *reinterpret_cast<const float4*>(w);
                    y[t][u] = *reinterpret_cast<const float4*>(w + 4);
                }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < RPT; ++u) {
                    v[u][0] += x[t][u].x; v[u][1] += x[t][u].y; v[u][2] += x[t][u].z; v[u][3] += x[t][u].w;
                    v[u][4] += y[t][u].x; v[u][5] += y[t][u].y; v[u][6] += y[t][u].z; v[u][7] += y[t][u].w;
                }
        }
        for (; s < nsplit; ++s) {
            float4 x[RPT], y[RPT];
#pragma unroll
            for (int u = 0; u < RPT; ++u) {
                const float* w = w0 + (size_t)s * slab + (size_t)(LANES * u) * d.N;
                x[u] = *reinterpret_cast<const float4*>(w);
                y[u] = *reinterpret_cast<const float4*>(w + 4);
            }
#pragma unroll
            for (int u = 0; u < RPT; ++u) {
                v[u][0] += x[u].x; v[u][1] += x[u].y; v[u][2] += x[u].z; v[u][3] += x[u].w;
                v[u][4] += y[u].x; v[u][5] += y[u].y; v[u][6] += y[u].z; v[u][7] += y[u].w;
            }
        }
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            if (fast) epi8_finish(e, row0 + rl + LANES * u, col, v[u], ops[u]);
            else epi_store8(e, row0 + rl + LANES * u, col, v[u]);   // v: the values before the fp16 store
#pragma unroll
            for (int i = 0; i < 8; ++i) fin[u][i] = (float)(f16)v[u][i];
        }
    }
    if (rl == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) pivot[co * 8 + i] = fin[0][i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float k = pivot[co * 8 + i];
        float S = 0.f, Q = 0.f;
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            const float a0 = fin[u][i] - k;
            S += a0;
            Q += a0 * a0;
        }
        red[(i * 2) * PLANE + rl * OCT + co] = S;
        red[(i * 2 + 1) * PLANE + rl * OCT + co] = Q;
    }
    __syncthreads();
    if (tid < COLS && blockIdx.x * COLS + tid < d.N) {
        float S = 0.f, Q = 0.f;
#pragma unroll 8
        for (int l = 0; l < LANES; ++l) {   // column tid = octet tid / 8, element tid % 8
            S += red[((tid & 7) * 2) * PLANE + l * OCT + (tid >> 3)];
            Q += red[((tid & 7) * 2 + 1) * PLANE + l * OCT + (tid >> 3)];
        }
        const float mean = pivot[tid] + S / 64.f, m2 = fmaxf(Q - S * S / 64.f, 0.f);
        reinterpret_cast<float2*>(d.out_stats)[(size_t)blockIdx.y * d.N + blockIdx.x * COLS + tid] = make_float2(mean, m2);
        if (d.stat_sums != nullptr)
            gn_sums_add(reinterpret_cast<unsigned long long*>(d.stat_sums), ((size_t)blockIdx.y * 64) / (size_t)d.stat_img_rows, d.N,
                        blockIdx.x * COLS + tid, mean, m2, 64);
    }
}

inline void launch_reduce_stats(const GemmArgs& a, int nsplit, hipStream_t stream) {
    const VdGemmDesc& d = a.d;
    // (128-column blocks -- 512-byte row pieces, twice the loads in flight -- measured SLOWER inside the forward, 10.98 vs 10.89 ms:
    // half as many blocks)
    hipLaunchKernelGGL(splitk_reduce_stats_kernel<8>, dim3((d.N + 63) / 64, d.M / 64, 1), dim3(256), 0, stream, a, nsplit);
}


}  // namespace

extern "C" size_t vd_gemm_workspace_bytes(const VdGemmDesc* d) {
    // upper bound for any split factor (split_k = 0) and tile shape the planner may choose; with arrival counters the slabs
    // are whole tiles (up to 256 x 320) in register order
    const size_t batch = d->batch > 0 ? d->batch : 1;
    const size_t ns = d->split_k > 0 ? (size_t)d->split_k : (size_t)VD_MAX_SPLIT_K;
    const size_t m = d->sync ? (size_t)d->M + 255 : (size_t)d->M, n = d->sync ? (size_t)d->N + 319 : (size_t)d->N;
    return batch * ns * m * n * sizeof(float);
}

namespace {
// Instantiation table.  w4b = 4 waves, ONE wave per SIMD (up to 512 registers), large per-wave tiles (gemm_big.hip).
enum TileCfg {
    T128x128 = 0, T128x64 = 1, T64x64 = 2, T128x128w8 = 3, T128x64w8 = 4, T256x128 = 5, T128x256 = 6, T128x320 = 7,
    T128x320b = 8, T128x256b = 9, T256x128b = 10, T128x160 = 11, T128x320b32 = 12, T128x128d = 13, T128x64d = 14, T64x64d = 15,
    // 32-deep K tiles, 4-stage ring (same LDS footprint as 64-deep / 2 stages, tiles issued 3 ahead instead of 1)
    T128x128q = 16, T128x64q = 17, T64x64q = 18, T128x128w8q = 19, T128x320q = 20, T128x160q = 21,
    T256x320 = 22, T256x320q = 23, T256x256 = 24,
    // 3 stages, mid-barrier main loop with pinned one-k-step-ahead fragment requests (gemm_kernel.h: MID)
    T256x128m = 25, T128x128m = 26,
    T_COUNT = 27
};
struct CfgInfo { int bm, bn; const char* name; };
const CfgInfo kCfg[T_COUNT] = {
    {128, 128, "gemm_f16_kernel<128,128,64,64,256,2,64>"},   {128, 64, "gemm_f16_kernel<128,64,64,32,256,2,64>"},
    {64, 64, "gemm_f16_kernel<64,64,32,32,256,2,64>"},       {128, 128, "gemm_f16_kernel<128,128,32,64,512,2,64>"},
    {128, 64, "gemm_f16_kernel<128,64,32,32,512,2,64>"},     {256, 128, "gemm_f16_kernel<256,128,64,64,512,2,64>"},
    {128, 256, "gemm_f16_kernel<128,256,64,64,512,2,64>"},   {128, 320, "gemm_f16_kernel<128,320,32,160,512,2,64>"},
    {128, 320, "gemm_f16_kernel<128,320,64,160,256,2,64>"},  {128, 256, "gemm_f16_kernel<128,256,64,128,256,3,64>"},
    {256, 128, "gemm_f16_kernel<256,128,128,64,256,3,64>"},  {128, 160, "gemm_f16_kernel<128,160,32,160,256,2,64>"},
    {128, 320, "gemm_f16_kernel<128,320,64,160,256,4,32>"},  {128, 128, "gemm_f16_kernel<128,128,64,64,256,3,64>"},
    {128, 64, "gemm_f16_kernel<128,64,64,32,256,3,64>"},     {64, 64, "gemm_f16_kernel<64,64,32,32,256,3,64>"},
    {128, 128, "gemm_f16_kernel<128,128,64,64,256,4,32>"},   {128, 64, "gemm_f16_kernel<128,64,64,32,256,4,32>"},
    {64, 64, "gemm_f16_kernel<64,64,32,32,256,4,32>"},       {128, 128, "gemm_f16_kernel<128,128,32,64,512,4,32>"},
    {128, 320, "gemm_f16_kernel<128,320,32,160,512,4,32>"},  {128, 160, "gemm_f16_kernel<128,160,32,160,256,4,32>"},
    {256, 320, "gemm_f16_kernel<256,320,64,160,512,2,64>"},  {256, 320, "gemm_f16_kernel<256,320,64,160,512,4,32>"},
    {256, 256, "gemm_f16_kernel<256,256,64,128,512,2,64>"},
    {256, 128, "gemm_f16_kernel<256,128,64,64,512,3,64>"},   {128, 128, "gemm_f16_kernel<128,128,32,64,512,3,64>"}};

// Tiles that are instantiated.  Rounds 1-5 carried 18 more as development tiles (256x128, 128x256, 128x160, the one-wave-per-SIMD
// "b" tiles, the 32-deep / 4-stage "q" ring, 256x320, 256x256, the mid-barrier "m" loop): each lost its in-forward A/B at least
// twice (profiles/HISTORY.md); round 6 removed the instantiations.  The table keeps their slots so configuration indices and
// names in old profiles stay meaningful.
bool cfg_built(int cfg) {
    switch (cfg) {
        case T128x128: case T128x64: case T64x64: case T128x128w8: case T128x64w8: case T128x320: case T128x128d: case T128x64d: case T64x64d:
            return true;
        default: return false;
    }
}

std::atomic<int> g_override{-1};

// Tuned launch table: (M, N, K, ksize, epilogue class) -> (tile configuration, split-K), filled by the host from a file
// measured INSIDE a UNet forward (tools/tune_forward.py -> lib/gemm_tune.py); shapes that are not in it go through the
// cost model below.  Read-mostly: writers take the mutex, the planner reads under it (a handful of entries).
struct TuneEntry { int M, N, K, ks, cls, cfg, nsplit; };
std::mutex g_tune_mu;
std::vector<TuneEntry> g_tune;
inline int epi_class(const VdGemmDesc& d) {
    return (d.act == VD_ACT_GEGLU ? 1 : 0) | ((d.flags & VD_EPI_LNFOLD) ? 2 : 0) | (d.a1 ? 4 : 0);
}

// which gemm_f16_kernel instances (as launched below) can emit VdGemmDesc.out_stats from their epilogue
bool cfg_emits_stats(int cfg) {
    switch (cfg) {
        case T128x128: return gemm_emits_stats<128, 128, 256, 2, 64, false>();
        case T128x64: return gemm_emits_stats<128, 64, 256, 2, 64, false>();
        case T64x64: return gemm_emits_stats<64, 64, 256, 2, 64, false>();
        case T128x128w8: return gemm_emits_stats<128, 128, 512, 2, 64, false>();
        case T128x64w8: return gemm_emits_stats<128, 64, 512, 2, 64, false>();
        case T128x320: return gemm_emits_stats<128, 320, 512, 2, 64, false>();
        case T128x128d: return gemm_emits_stats<128, 128, 256, 3, 64, false>();
        case T128x64d: return gemm_emits_stats<128, 64, 256, 3, 64, false>();
        case T64x64d: return gemm_emits_stats<64, 64, 256, 3, 64, false>();
        default: return false;   // development tiles: the host falls back to vd_chan_stats_f16
    }
}

// rows per statistics partial the planned launch writes to d.out_stats (0: it cannot)
int plan_stat_rows(const GemmArgs& a, int cfg, int nsplit, const ConvHaloArgs* halo) {
    const VdGemmDesc& d = a.d;
    if (d.batch != 1 || (d.flags & (VD_EPI_OUT_F32 | VD_EPI_LNFOLD | VD_EPI_BIAS_ALONG_M)) || d.act == VD_ACT_GEGLU) return 0;
    if (d.N % 8 != 0 || d.ldc % 8 != 0 || ((d.flags & VD_EPI_RESIDUAL) && d.ldr % 8 != 0)) return 0;
    const int HW = d.stat_img_rows;
    if (HW <= 0 || d.M % HW != 0) return 0;
    if (cfg >= T_COUNT) {
        // the halo conv's epilogue runs in the kernel itself when it is not split, or when the split is reduced by the tile's
        // last block (ticket counters supplied and enough of them: the condition of vd_gemm_f16)
        const bool in_kernel = nsplit <= 1 || (d.sync != nullptr && halo != nullptr && 2l * halo->g.tiles_m * halo->g.tiles_n <= VD_GEMM_SYNC_INTS);
        if (in_kernel) {
            if (halo == nullptr || HW != halo->Hv * halo->Wv) return 0;
            return halo->ngrp == 1 ? d.M / halo->g.tiles_m : HW;
        }
        return (d.sync == nullptr && HW % 64 == 0) ? 64 : 0;
    }
    if (nsplit > 1) return (d.sync == nullptr && HW % 64 == 0) ? 64 : 0;   // splitk_reduce_stats_kernel
    if (!cfg_emits_stats(cfg)) return 0;
    const int bm = kCfg[cfg].bm;
    return HW % bm == 0 ? bm : (bm % HW == 0 ? HW : 0);
}

// validate + normalise the descriptor and pick tile shape / split factor
int plan_gemm(const VdGemmDesc* dp, GemmArgs& a, int& cfg_out, int& nsplit_out, ConvHaloArgs* halo = nullptr) {
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
    if (d.stat_img_rows <= 0) d.stat_img_rows = d.Hout * d.Wout;
    a.stat_rows = 0;
    a.hoist = 1;   // epilogue operands are requested in front of the main loop (gemm_kernel.h: HOIST)
#ifdef VD_TIMELINE
    a.tl = nullptr;
#endif
    if (d.skip_a0 != nullptr || d.skip_w != nullptr) {
        VD_REQUIRE(d.skip_a0 && d.skip_w && d.ksize == 3, "vd_gemm_f16: the folded skip convolution needs skip_a0, skip_w and a 3x3 convolution");
        if (d.skip_a1 == nullptr) d.skip_c1 = 0;
        if (d.skip_lda0 <= 0) d.skip_lda0 = d.skip_c0;
        if (d.skip_lda1 <= 0) d.skip_lda1 = d.skip_c1;
        if (d.skip_ldw <= 0) d.skip_ldw = d.skip_c0 + d.skip_c1;
    }
    const int n_out = (d.act == VD_ACT_GEGLU) ? d.N / 2 : d.N;
    if (d.ldc <= 0) d.ldc = n_out;
    if (d.ldr <= 0) d.ldr = n_out;
    // the epilogues read residual / row-vector / LayerNorm-statistics through buffer descriptors with 32-bit byte offsets
    VD_REQUIRE(!(d.flags & VD_EPI_RESIDUAL) || (unsigned long long)d.M * (unsigned)d.ldr < (1ull << 30),
               "vd_gemm_f16: residual larger than 2 GiB per batch entry (32-bit buffer offsets); split the rows");
    VD_REQUIRE(!(d.flags & VD_EPI_LNFOLD) || (unsigned long long)d.M * (unsigned)(d.batch > 0 ? d.batch : 1) < (1ull << 27),
               "vd_gemm_f16: LayerNorm fold over more than 2^27 rows (32-bit buffer offsets); split the rows");
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
    const bool lnfold = (d.flags & VD_EPI_LNFOLD) != 0;
    if (lnfold) {
        VD_REQUIRE(d.colsum != nullptr, "vd_gemm_f16: LayerNorm fold needs colsum");
        // statistics inside the K loop (one-pass E[x^2] - mean^2 on the raw fp16 operands, less robust than the two-pass
        // vd_row_stats_f16) only on explicit request: a forgotten ln_stats pointer is an error, not a silent downgrade
        VD_REQUIRE(d.ln_stats != nullptr || (d.flags & VD_EPI_LN_INLOOP),
                   "vd_gemm_f16: LayerNorm fold needs ln_stats (or VD_EPI_LN_INLOOP for in-loop statistics)");
        VD_REQUIRE(d.ln_stats == nullptr || !(d.flags & VD_EPI_LN_INLOOP), "vd_gemm_f16: VD_EPI_LN_INLOOP takes no ln_stats");
        VD_REQUIRE(d.ksize == 1 && d.a1 == nullptr && d.split_k <= 1 && !(d.flags & (VD_EPI_OUT_F32 | VD_EPI_BIAS_ALONG_M)),
                   "vd_gemm_f16: LayerNorm fold takes a plain single-source A, fp16 output, no split-K");
        VD_REQUIRE(((size_t)d.colsum & 15) == 0 && ((size_t)d.ln_stats & 7) == 0, "vd_gemm_f16: colsum must be 16-byte, ln_stats 8-byte aligned");
    }

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
    struct Cand { TileCfg cfg; int cap; float t_solo, t_full, lo, hi, fix; bool split_ok; };
    static const Cand cands[] = {
        {T128x128, 512, 0.75f, 1.10f, 0.50f, 1.00f, 6.5f, true},
        {T128x64, 768, 0.50f, 1.00f, 0.25f, 0.83f, 4.f, true},
        {T64x64, 1024, 0.38f, 0.78f, 0.25f, 0.75f, 3.f, true},
        {T128x320, 256, 1.20f, 1.50f, 0.50f, 1.00f, 6.f, false}};
    const int zb = d.batch;
    const bool can_split = (d.ws != nullptr || d.split_k > 1) && d.act != VD_ACT_GEGLU && !(d.flags & VD_EPI_OUT_F32) && !lnfold;
    auto model_us = [&](const Cand& c, int ns) {
        const int bm = kCfg[c.cfg].bm, bn = kCfg[c.cfg].bn;
        const int tiles = ((d.M + bm - 1) / bm) * ((d.N + bn - 1) / bn) * zb;
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
    // 3x3 convolutions on patch-shaped output grids: the halo-resident kernel (conv_halo.hip), unless a GEMM tile is forced
    static const bool tile_env = getenv("VD_GEMM_TILE") != nullptr;
    bool tuned_gemm_tile = false;   // a tuned-table entry for this 3x3 problem pins a gemm_f16_kernel tile: it vetoes the halo path
    if (d.ksize == 3) {
        std::lock_guard<std::mutex> lk(g_tune_mu);
        const int cls = epi_class(d);
        for (const TuneEntry& t : g_tune)
            if (t.M == d.M && t.N == d.N && t.K == d.K && t.ks == 3 && t.cls == cls && d.batch == 1) { tuned_gemm_tile = true; break; }
    }
    if (d.ksize == 3 && g_override.load(std::memory_order_relaxed) < 0 && !tile_env && !tuned_gemm_tile) {
        ConvHaloArgs local;
        ConvHaloArgs* h = halo ? halo : &local;
        int hv = 0, hns = 1;
        if (vd_conv_halo_plan(&a, can_split ? 1 : 0, h, &hv, &hns)) {
            if (hns > 1) {
                VD_REQUIRE(d.ws != nullptr, "vd_gemm_f16: split_k=%d needs a workspace", hns);
                VD_REQUIRE(hns <= VD_MAX_SPLIT_K, "vd_gemm_f16: split_k=%d > %d", hns, VD_MAX_SPLIT_K);
            }
            a.tiles_m = h->g.tiles_m;
            a.tiles_n = h->g.tiles_n;
            a.kt_per_split = a.kt_total;
            cfg_out = T_COUNT + hv;
            nsplit_out = hns;
            a.stat_rows = h->g.stat_rows = plan_stat_rows(a, cfg_out, hns, h);
            return VD_OK;
        }
    }
    VD_REQUIRE(d.skip_a0 == nullptr, "vd_gemm_f16: the folded skip convolution is taken by the halo-resident 3x3 convolution only (vd_gemm_skip_ok)");
    TileCfg cfg = T64x64;
    int nsplit = 1;
    if (d.act == VD_ACT_GEGLU) {
        // 8 waves: the erf-heavy epilogue of one wave overlaps MFMAs of others.  (The 256x256 tile, 64x128 per wave, is
        // 1 % slower over the forward for the M >= 4096 GEGLU projections: 13.48 vs 13.36 ms, same box.)
        cfg = T128x128w8;
    }
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
    } else if (a.kt_total < 20 && d.M >= 4096 && d.N <= 640) {
        // short-K projections of the high-resolution levels (K = 320 .. 1280 against 8192+ rows): output-write bound and
        // short-lived blocks -- 128x64 tiles on 8 waves (4 waves per SIMD, 4+ blocks per CU at different phases) beat the
        // large tiles by 7-24 % inside a forward.  Round 5 (re-measured with the repaired epilogues, tools/unet_forward.py with
        // VD_FWD_TUNE pins, same box): at N = 640 and M >= 8192 the 128x128 tile on 8 waves is 6 us per launch faster than the
        // 128x64 one (20 launches per forward: 10.605 -> 10.48-10.50 ms); at N = 320 both tiles measure the same
        // and at N = 640 with 4096 <= M < 8192 (the 32x32 level at CFG batch 4, BASELINE configs[3]) the 128x64 tile on 4 waves with three
        // stages wins 0.06 ms per dual-context forward over the 8-wave one (tools/tune_graph.py --workload dual)
        cfg = (d.N > 320 && d.M >= 8192) ? T128x128w8 : (d.N > 320 ? T128x64d : T128x64w8);
    } else {
        float best = 1e30f;
        const int ns_max = (d.split_k > 0) ? d.split_k : ((can_split && a.kt_total >= 32) ? VD_MAX_SPLIT_K / 2 : 1);
        for (const Cand& c : cands)
            for (int ns = (d.split_k > 0 ? d.split_k : 1); ns <= ns_max; ++ns) {
                if (ns > 1 && (a.kt_total / ns < 8 || !c.split_ok)) break;
                const float t = model_us(c, ns);
                if (t < best) { best = t; cfg = c.cfg; nsplit = ns; }
            }
        // the same N = 640 rows with a deep K (FF-out of the 32x32 level, K = 2560): 8 waves on the 128x128 tile, 0.02-0.04 ms
        // per forward over the 4-wave one (tools/tune_graph.py, two sessions)
        if (cfg == T128x128 && nsplit == 1 && d.ksize <= 1 && d.N > 320 && d.N <= 640 && d.M >= 8192) cfg = T128x128w8;
    }
    bool tuned = false;
    {   // also with a caller-fixed split (ops.gemm plans first, then launches with split_k = the planned factor): the tuned
        // tile is honoured when its split factor is the one requested
        std::lock_guard<std::mutex> lk(g_tune_mu);
        const int cls = epi_class(d);
        for (const TuneEntry& t : g_tune)
            if (t.M == d.M && t.N == d.N && t.K == d.K && t.ks == d.ksize && t.cls == cls && d.batch == 1) {
                if (d.split_k > 0 && (t.nsplit > 0 ? t.nsplit : 1) != d.split_k) break;
                if (t.nsplit > 1 && !can_split) break;
                if (d.act == VD_ACT_GEGLU && kCfg[t.cfg].bn % 128 != 0) break;
                cfg = (TileCfg)t.cfg;
                nsplit = t.nsplit > 0 ? t.nsplit : 1;
                tuned = true;
                break;
            }
    }
    {   // developer override (vd_gemm_set_override / VD_GEMM_TILE=<n>): never set in production runs
        static const char* ov_env = getenv("VD_GEMM_TILE");
        int ov = g_override.load(std::memory_order_relaxed);
        if (ov < 0 && ov_env) ov = atoi(ov_env);
        const bool geglu_ok = ov == T128x128 || ov == T128x128w8 || ov == T128x128d;
        if (ov >= 0 && ov < T_COUNT && cfg_built(ov) && (d.act != VD_ACT_GEGLU || geglu_ok) && !(d.M < 96 || d.N < 96)) {
            cfg = (TileCfg)ov;
            // re-plan the split for the forced tile: fill the chip once (one block per CU for the 1-block-per-CU tiles)
            nsplit = 1;
            if (d.split_k <= 0 && can_split && a.kt_total >= 32) {
                const int tiles = ((d.M + kCfg[cfg].bm - 1) / kCfg[cfg].bm) * ((d.N + kCfg[cfg].bn - 1) / kCfg[cfg].bn) * zb;
                const int slots = (cfg == T128x128 || cfg == T128x128d) ? 512
                                  : (cfg == T128x64 || cfg == T128x64d) ? 768
                                  : (cfg == T64x64 || cfg == T64x64d) ? 1024 : 256;
                nsplit = slots / tiles;
                if (nsplit < 1) nsplit = 1;
                if (nsplit > VD_MAX_SPLIT_K / 2) nsplit = VD_MAX_SPLIT_K / 2;
                while (nsplit > 1 && a.kt_total / nsplit < 8) --nsplit;
            }
        }
    }
    if (d.split_k > 0) nsplit = d.split_k;
    if (nsplit > a.kt_total) nsplit = a.kt_total;
    if (g_override.load(std::memory_order_relaxed) < 0 && !tuned) {
        const int ktps = (a.kt_total + nsplit - 1) / nsplit;
        // grids that cannot fill the CUs twice over are latency-bound per block: those get a deeper ring of 64-deep tiles
        // (round 4, tools/probes/gemm_timeline.py: raising the limit to "every block still co-resident" -- 768 blocks of 64x64,
        // 512 of 128x64 -- shortens the K loop of the M = 2048, N = K = 1280 projections from 11.6 to 10.1 us per block, not to
        // half: at 640 blocks x 16 KiB per k-step the loop runs at the L2 -> LDS fill rate, not at one round trip per k-step.
        // Forward unchanged (10.57 vs 10.57 ms), so the limit stays; VD_GEMM_DEEP_MAX moves it.)
        const long grid_blocks = (long)((d.M + kCfg[cfg].bm - 1) / kCfg[cfg].bm) * ((d.N + kCfg[cfg].bn - 1) / kCfg[cfg].bn) * nsplit * zb;
        const long deep64 = 400, deep128 = 400;
        if (ktps >= 8) {
            if (cfg == T128x64 && grid_blocks <= deep128) cfg = T128x64d;
            else if (cfg == T64x64 && grid_blocks <= deep64) cfg = T64x64d;
            else if (cfg == T64x64 && nsplit == 1 && ktps >= 16 && d.M % 128 == 0 && grid_blocks / 2 <= deep128 && d.act != VD_ACT_GEGLU) {
                // round 5: a 64x64 grid too large for the deep ring whose 128x64 grid fits it (M = 2048, N = K = 1280: 640 -> 320
                // blocks, all co-resident with three stages): 10.166 -> 10.120 ms per forward for the 20 C x C launches of the
                // 16x16 level (VD_FWD_TUNE pin A/B)
                cfg = T128x64d;
            }
        }
    }
    // LayerNorm-folded q|k|v of the 32x32 level (K = 640, N = 1920): the 128x128 tile on 8 waves, 0.015 (t2i) .. 0.05 ms (i2v) per
    // forward over the 4-wave one in all four workloads (tools/tune_graph.py)
    if (lnfold && !tuned && g_override.load(std::memory_order_relaxed) < 0 && (cfg == T128x128 || cfg == T128x320) && nsplit == 1 && a.kt_total > 5 && a.kt_total <= 10 && d.M >= 4096 &&
        d.act != VD_ACT_GEGLU)
        cfg = T128x128w8;
    if (lnfold) {
        // the LayerNorm fold is a compile-time variant of the kernel, instantiated for these tiles only
        switch (cfg) {
            case T128x128: case T128x64: case T64x64: case T128x128w8: case T128x64w8: case T128x320: case T64x64d: break;
            case T128x64d: cfg = T128x64; break;
            case T128x128d: cfg = T128x128; break;
            default: cfg = (d.act == VD_ACT_GEGLU) ? T128x128w8 : T128x128; break;
        }
    }
    const int bm = kCfg[cfg].bm, bn = kCfg[cfg].bn;
    a.tiles_m = (d.M + bm - 1) / bm;
    a.tiles_n = (d.N + bn - 1) / bn;
    if (nsplit > a.kt_total) nsplit = a.kt_total;
    if (nsplit > 1) {
        VD_REQUIRE(d.ws != nullptr, "vd_gemm_f16: split_k=%d needs a workspace", nsplit);
        VD_REQUIRE(d.act != VD_ACT_GEGLU && !lnfold, "vd_gemm_f16: split-K with GEGLU / LayerNorm-fold epilogue unsupported");
        VD_REQUIRE(nsplit <= VD_MAX_SPLIT_K, "vd_gemm_f16: split_k=%d > %d", nsplit, VD_MAX_SPLIT_K);
    }
    a.kt_per_split = (a.kt_total + nsplit - 1) / nsplit;
    nsplit = (a.kt_total + a.kt_per_split - 1) / a.kt_per_split;
    cfg_out = (int)cfg;
    nsplit_out = nsplit;
    a.stat_rows = plan_stat_rows(a, cfg_out, nsplit, nullptr);
    return VD_OK;
}
}  // namespace

extern "C" int vd_gemm_stat_rows(const VdGemmDesc* dp, int* rows) {
    GemmArgs a;
    ConvHaloArgs halo;
    int c = 0, n = 1;
    const int rc = plan_gemm(dp, a, c, n, &halo);
    if (rc != VD_OK) return rc;
    // the split-K in-kernel fix-up is decided at launch (vd_gemm_f16 drops d.sync when the tiles outnumber the counters);
    // plan_stat_rows already answers 0 whenever the caller supplied counters
    if (rows) *rows = a.stat_rows;
    return VD_OK;
}

extern "C" int vd_gemm_skip_ok(const VdGemmDesc* dp) {
    if (dp == nullptr || dp->skip_a0 == nullptr || dp->skip_w == nullptr) return 0;
    GemmArgs a;
    ConvHaloArgs halo;
    int c = 0, n = 1;
    if (plan_gemm(dp, a, c, n, &halo) != VD_OK) return 0;
    return c == T_COUNT + 12 ? 1 : 0;
}

// launches whose part 2 can accumulate VdGemmDesc.row_sums: an unsplit gemm_f16_kernel (not the halo conv, not a reduce kernel)
// with the fast write-out path (vector-aligned fp16 output, residual OR row vector) and a power-of-two number of 16-byte
// segments per tile row
static bool row_sums_ok(const GemmArgs& a, int cfg, int nsplit) {
    const VdGemmDesc& d = a.d;
    if (cfg >= T_COUNT || nsplit > 1 || d.act == VD_ACT_GEGLU) return false;
    if (d.flags & (VD_EPI_OUT_F32 | VD_EPI_BIAS_ALONG_M)) return false;
    if ((d.flags & VD_EPI_RESIDUAL) && (d.flags & VD_EPI_ROWVEC)) return false;
    if ((d.N & 7) || (d.ldc & 7) || (d.ldr & 7)) return false;
    const int ch = kCfg[cfg].bn / 8;
    if ((ch & (ch - 1)) != 0 || ch > 64) return false;
    // two-pass epilogues (256 x 320) and the one-wave-per-SIMD development tiles keep the plain path
    return cfg == T128x128 || cfg == T128x64 || cfg == T64x64 || cfg == T128x128w8 || cfg == T128x64w8 || cfg == T128x128d ||
           cfg == T128x64d || cfg == T64x64d || cfg == T128x256 || cfg == T256x128;
}

extern "C" int vd_gemm_row_sums_ok(const VdGemmDesc* dp) {
    if (dp == nullptr) return 0;
    GemmArgs a;
    int c = 0, n = 1;
    if (plan_gemm(dp, a, c, n) != VD_OK) return 0;
    return row_sums_ok(a, c, n) ? 1 : 0;
}

extern "C" int vd_gemm_plan(const VdGemmDesc* dp, int* tile_cfg, int* nsplit) {
    GemmArgs a;
    int c = 0, n = 1;
    const int rc = plan_gemm(dp, a, c, n);
    if (rc != VD_OK) return rc;
    if (tile_cfg) *tile_cfg = c;
    if (nsplit) *nsplit = n;
    return VD_OK;
}

extern "C" const char* vd_gemm_config_name(int tile_cfg) {
    if (tile_cfg >= T_COUNT) return vd_conv_halo_name(tile_cfg - T_COUNT);
    return (tile_cfg >= 0 && tile_cfg < T_COUNT) ? kCfg[tile_cfg].name : nullptr;
}
extern "C" int vd_gemm_num_configs(void) { return T_COUNT; }
extern "C" int vd_gemm_tune_set(int M, int N, int K, int ksize, int epi_cls, int tile_cfg, int nsplit) {
    VD_REQUIRE(tile_cfg >= 0 && tile_cfg < T_COUNT && nsplit >= 0 && nsplit <= VD_MAX_SPLIT_K, "vd_gemm_tune_set: bad entry");
    std::lock_guard<std::mutex> lk(g_tune_mu);
    for (TuneEntry& t : g_tune)
        if (t.M == M && t.N == N && t.K == K && t.ks == (ksize > 0 ? ksize : 1) && t.cls == epi_cls) {
            t.cfg = tile_cfg;
            t.nsplit = nsplit;
            return VD_OK;
        }
    g_tune.push_back({M, N, K, ksize > 0 ? ksize : 1, epi_cls, tile_cfg, nsplit});
    return VD_OK;
}
extern "C" int vd_gemm_tune_clear(void) {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    g_tune.clear();
    return VD_OK;
}
extern "C" int vd_gemm_set_override(int tile_cfg) {
    if (tile_cfg >= 0 && !cfg_built(tile_cfg)) {
        vd_set_error("vd_gemm_set_override: tile configuration %d is not instantiated", tile_cfg);
        return VD_ERR_UNSUPPORTED;
    }
    g_override.store(tile_cfg, std::memory_order_relaxed);
    return VD_OK;
}

#ifdef VD_TIMELINE
static unsigned long long* g_timeline = nullptr;
// development build only: per-block phase stamps of the next gemm_f16_kernel launches go to buf (8 x u64 per block), null = off
extern "C" void vd_debug_set_timeline(void* buf) { g_timeline = reinterpret_cast<unsigned long long*>(buf); }
#endif

extern "C" int vd_gemm_f16(const VdGemmDesc* dp, hipStream_t stream) {
    GemmArgs a;
    ConvHaloArgs halo;
    int cfg = 0, nsplit = 1;
    int rc = plan_gemm(dp, a, cfg, nsplit, &halo);
    if (rc != VD_OK) return rc;
#ifdef VD_TIMELINE
    a.tl = g_timeline;
    halo.g.tl = g_timeline;
#endif
    const VdGemmDesc& d = a.d;
    if (d.out_stats != nullptr && a.stat_rows == 0) {
        vd_set_error("vd_gemm_f16: out_stats requested but the planned launch cannot emit statistics (vd_gemm_stat_rows = 0)");
        return VD_ERR_UNSUPPORTED;
    }
    VD_REQUIRE(((size_t)d.out_stats & 7) == 0, "vd_gemm_f16: out_stats must be 8-byte aligned");
    VD_REQUIRE(d.stat_sums == nullptr || (d.out_stats != nullptr && ((size_t)d.stat_sums & 7) == 0 && d.stat_img_rows > 0 && d.M % d.stat_img_rows == 0),
               "vd_gemm_f16: stat_sums rides on out_stats (8-byte aligned, whole images of stat_img_rows rows)");
    if (d.row_sums != nullptr && !row_sums_ok(a, cfg, nsplit)) {
        vd_set_error("vd_gemm_f16: row_sums requested but the planned launch cannot accumulate them (vd_gemm_row_sums_ok)");
        return VD_ERR_UNSUPPORTED;
    }
    if (d.flags & VD_EPI_LN_SUMS) VD_REQUIRE((d.flags & VD_EPI_LNFOLD) && d.ln_stats != nullptr, "vd_gemm_f16: VD_EPI_LN_SUMS needs VD_EPI_LNFOLD and ln_stats");
    // (VD_EPI_GROUPNORM of ABI 5-7 -- the consuming GroupNorm inside the kernel that sums split-K slabs -- measured neutral and was
    // removed with ABI 8: the flag bits and the gn_* descriptor fields are reserved)
    VD_REQUIRE(!(d.flags & (VD_EPI_GROUPNORM | VD_EPI_GN_SILU)), "vd_gemm_f16: VD_EPI_GROUPNORM was removed in ABI 8 (run the GroupNorm from out_stats)");
    const int zb = d.batch;
    a.nt_store = 1;   // non-temporal stores of write-once outputs
    {   // which operand an XCD keeps in its private L2 (see the tile mapping in gemm_kernel.h): fabric bytes of the two orders
        const int tiles_m = cfg >= T_COUNT ? halo.g.tiles_m : a.tiles_m, tiles_n = cfg >= T_COUNT ? halo.g.tiles_n : a.tiles_n;
        const double wbytes = (double)d.N * d.K * 2.0, abytes = (double)a.a0_bytes + a.a1_bytes;
        const double runs = (double)tiles_m * tiles_n / 8.0;   // tiles per XCD
        // n fastest: an XCD covers min(tiles_n, runs) column panels and runs / tiles_n (>= 1) row panels
        const double nfast = 8.0 * wbytes * (runs < tiles_n ? runs / tiles_n : 1.0) + abytes * (runs < tiles_n ? tiles_n / runs : 1.0);
        const double mfast = 8.0 * abytes * (runs < tiles_m ? runs / tiles_m : 1.0) + wbytes * (runs < tiles_m ? tiles_m / runs : 1.0);
        a.mfast = (zb == 1 && mfast < 0.8 * nfast) ? 1 : 0;
        halo.g.mfast = a.mfast;
    }
    if (cfg >= T_COUNT) {   // halo-resident 3x3 convolution; split-K slabs go through the same reduce kernel
        halo.g.nt_store = a.nt_store;
        // (the channel-chunk split is reduced by the reduce launch: the ticketed in-kernel reduction of rounds 4-5 -- VD_HALO_FIXUP,
        // with or without the XCD-local exchange -- measured equal at a 2-way split and 9 us slower per conv at 4-way; removed in round 6)
        halo.g.d.sync = nullptr;
        halo.g.xcd_local = 0;
        rc = vd_conv_halo_launch(&halo, cfg - T_COUNT, nsplit, stream);
        if (rc != VD_OK) return rc;
        if (nsplit > 1 && halo.g.d.sync == nullptr) {   // no ticket counters: slabs + the reduce kernel
            a.d.sync = nullptr;
            if (d.out_stats != nullptr) {
                launch_reduce_stats(a, nsplit, stream);
                return vd_check_launch("vd_gemm_f16/splitk_reduce_stats");
            }
            const size_t total = (size_t)d.M * ((d.N + 7) / 8);
            int blocks = (int)((total + 255) / 256);
            if (blocks > 4096) blocks = 4096;
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks, 1, 1), dim3(256), 0, stream, a, nsplit);
            return vd_check_launch("vd_gemm_f16/splitk_reduce");
        }
        return VD_OK;
    }
    if ((long)a.tiles_m * a.tiles_n * zb > VD_GEMM_SYNC_INTS) a.d.sync = nullptr;  // more tiles than counters: two-kernel path
    if (d.flags & VD_EPI_LNFOLD) {
        switch (cfg) {
            case T128x128: rc = launch_cfg<128, 128, 64, 64, 256, 2, 64, 2, true>(a, nsplit, stream); break;
            case T128x64: rc = launch_cfg<128, 64, 64, 32, 256, 2, 64, 2, true>(a, nsplit, stream); break;
            case T64x64: rc = launch_cfg<64, 64, 32, 32, 256, 2, 64, 2, true>(a, nsplit, stream); break;
            case T64x64d: rc = launch_cfg<64, 64, 32, 32, 256, 3, 64, 2, true>(a, nsplit, stream); break;
            case T128x128w8: rc = launch_cfg<128, 128, 32, 64, 512, 2, 64, 4, true>(a, nsplit, stream); break;
            case T128x64w8: rc = launch_cfg<128, 64, 32, 32, 512, 2, 64, 4, true>(a, nsplit, stream); break;
            case T128x320: rc = launch_cfg<128, 320, 32, 160, 512, 2, 64, 2, true>(a, nsplit, stream); break;
            default:
                vd_set_error("vd_gemm_f16: tile configuration %d has no LayerNorm-fold instance", cfg);
                return VD_ERR_UNSUPPORTED;
        }
        return rc;
    }
    switch (cfg) {
        case T128x128: rc = launch_cfg<128, 128, 64, 64, 256, 2, 64, 2>(a, nsplit, stream); break;
        case T128x64: rc = launch_cfg<128, 64, 64, 32, 256, 2, 64, 2>(a, nsplit, stream); break;
        case T64x64: rc = launch_cfg<64, 64, 32, 32, 256, 2, 64, 2>(a, nsplit, stream); break;
        case T128x128w8: rc = launch_cfg<128, 128, 32, 64, 512, 2, 64, 4>(a, nsplit, stream); break;
        case T128x64w8: rc = launch_cfg<128, 64, 32, 32, 512, 2, 64, 4>(a, nsplit, stream); break;
        case T128x320: rc = launch_cfg<128, 320, 32, 160, 512, 2, 64, 2>(a, nsplit, stream); break;
        case T128x128d: rc = launch_cfg<128, 128, 64, 64, 256, 3, 64, 1>(a, nsplit, stream); break;
        case T128x64d: rc = launch_cfg<128, 64, 64, 32, 256, 3, 64, 2>(a, nsplit, stream); break;
        case T64x64d: rc = launch_cfg<64, 64, 32, 32, 256, 3, 64, 2>(a, nsplit, stream); break;
        default:
            vd_set_error("vd_gemm_f16: tile configuration %d is not instantiated", cfg);
            return VD_ERR_UNSUPPORTED;
    }
    if (rc != VD_OK) return rc;
    if (nsplit > 1 && a.d.sync == nullptr) {
        if (d.out_stats != nullptr) {   // plan_stat_rows: batch 1, fp16 output, whole 64-row blocks per image
            launch_reduce_stats(a, nsplit, stream);
            return vd_check_launch("vd_gemm_f16/splitk_reduce_stats");
        }
        const size_t total = (size_t)d.M * ((d.N + 7) / 8);
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks, 1, zb), dim3(256), 0, stream, a, nsplit);
        return vd_check_launch("vd_gemm_f16/splitk_reduce");
    }
    return VD_OK;
}


// ---- helpers for conv_wstream.hip (its own translation unit: other compiler flags) ---------------------------------------
// validate + normalise `desc` exactly as vd_gemm_f16 does; gemm_args_out: a GemmArgs
int vd_gemm_normalise(const VdGemmDesc* desc, void* gemm_args_out) {
    int cfg = 0, ns = 1;
    return plan_gemm(desc, *static_cast<GemmArgs*>(gemm_args_out), cfg, ns);
}
// sum `nsplit` fp32 slabs of d.ws and run the fused epilogue of the (normalised) descriptor; with d.out_stats also the
// per-channel statistics in partials of 64 rows
int vd_gemm_launch_reduce(const void* gemm_args, int nsplit, hipStream_t stream) {
    const GemmArgs& a = *static_cast<const GemmArgs*>(gemm_args);
    const VdGemmDesc& d = a.d;
    if (d.out_stats != nullptr) {
        launch_reduce_stats(a, nsplit, stream);
        return vd_check_launch("splitk_reduce_stats");
    }
    const size_t total = (size_t)d.M * ((d.N + 7) / 8);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks, 1, 1), dim3(256), 0, stream, a, nsplit);
    return vd_check_launch("splitk_reduce");
}
