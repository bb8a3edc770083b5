// vd_conv3x3_wstream_f16: launcher and instances of conv3x3_wstream_kernel (conv_wstream_kernel.h).  Built WITHOUT
// -amdgpu-mfma-vgpr-form: the 128 accumulator registers of a wave live in AGPRs, the architectural VGPRs hold the ring of
// weight fragments in flight, the pixel fragments and the addresses.
#include "conv_wstream_kernel.h"
#include "conv_wsk_kernel.h"
#include "gemm_wstream_kernel.h"

// gemm.hip
int vd_gemm_normalise(const VdGemmDesc* desc, void* gemm_args_out);
int vd_gemm_launch_reduce(const void* gemm_args, int nsplit, hipStream_t stream);

namespace {
// geometry the kernel takes: 3x3 / stride 1 / pad 1 on 8x8 images, an even number of images, 64-channel chunks, N % 256 == 0
const char* wstream_reject(const VdGemmDesc& d) {
    if (d.ksize != 3 || d.stride != 1 || d.pad != 1 || d.ups != 0 || d.batch != 1) return "3x3 / stride 1 / pad 1, no upsample, batch 1";
    if (d.Hin != 8 || d.Win != 8 || d.Hout != 8 || d.Wout != 8) return "8x8 images";
    if (d.M % 128 != 0 || d.N % 256 != 0) return "an even number of images and N % 256 == 0";
    if (d.c0 % 64 != 0 || d.c1 % 64 != 0) return "channel counts in multiples of 64";
    if ((d.flags & (VD_EPI_LNFOLD | VD_EPI_OUT_F32 | VD_EPI_BIAS_ALONG_M)) || d.act == VD_ACT_GEGLU) return "a plain fp16 epilogue";
    return nullptr;
}

std::atomic<int> g_ws_variant{-1};   // -1: not read from the environment yet
std::atomic<int> g_ws_blocks{-1};

template <int D, int OCC, int IPB = 2>
int launch_wstream(const WsArgs& w, int blocks, hipStream_t stream) {
    constexpr int LDS = 2 * 4 * (((IPB * WS_GPX * 128 + 1023) / 1024 + 3) / 4) * 1024;   // two halo buffers
    if constexpr (LDS > 64 * 1024) {
        static std::atomic<unsigned long long> done{0};
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(done.load(std::memory_order_acquire) & bit)) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wstream_kernel<D, OCC, IPB>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            if (e != hipSuccess) {
                vd_set_error("vd_conv3x3_wstream_f16: cannot reserve %d bytes of LDS: %s", LDS, hipGetErrorString(e));
                return VD_ERR_LAUNCH;
            }
            done.fetch_or(bit, std::memory_order_release);
        }
    }
    hipLaunchKernelGGL((conv3x3_wstream_kernel<D, OCC, IPB>), dim3(blocks), dim3(256), LDS, stream, w);
    return vd_check_launch("vd_conv3x3_wstream_f16");
}
// images per block of the split kernel: 2.  (Round 5 also built a 4-images-per-block geometry -- a wave owns 32 channels over 256
// pixels, every weight byte enters half as many CUs -- to test whether the launch is bound by what a CU ingests: it is not, 33.7 vs
// 32.0 us per launch in isolation, 10.44 / 10.47 vs 10.41 ms per forward (profiles/HISTORY.md); removed in round 6.)
constexpr int wstream_ipb(const VdGemmDesc&) { return 2; }
}  // namespace

// Development hook (tests, A/B runs): instance 0 = 12 k-steps of weights in flight at one wave per SIMD (default), 1 = 9 at
// one wave, 2 / 3 = 4 / 6 at two waves per SIMD; target_blocks = grid size the split over chunks aims for (256).
extern "C" int vd_conv3x3_wstream_set_variant(int variant, int target_blocks) {
    VD_REQUIRE(variant >= 0 && variant <= 3 && target_blocks > 0, "vd_conv3x3_wstream_set_variant: bad arguments");
    g_ws_variant.store(variant, std::memory_order_relaxed);
    g_ws_blocks.store(target_blocks, std::memory_order_relaxed);
    return VD_OK;
}

extern "C" int vd_conv3x3_wstream_supported(const VdGemmDesc* dp) {
    if (dp == nullptr) return 0;
    VdGemmDesc d = *dp;
    if (d.c0 <= 0) d.c0 = d.K / 9;
    if (d.a1 == nullptr) d.c1 = 0;
    if (d.batch <= 0) d.batch = 1;
    if (d.stride <= 0) d.stride = 1;
    return wstream_reject(d) == nullptr && d.K == 9 * (d.c0 + d.c1) ? 1 : 0;
}

namespace {
// the whole-K kernel (conv_wsk_kernel.h) takes the launch when its 128-pixel x 32-channel tiles fill at least half the chip, the
// chunk sequence fits its unrolled loop, there is no folded skip convolution and no GroupNorm in the (absent) reduce
bool wsk_takes(const VdGemmDesc& d, int nchunks, int nskip) {
    const char* env = getenv("VD_WSK");   // development switch: 0 = always the split kernel + reduce launch (read per call: tests flip it)
    if (env && env[0] == '0') return false;
    const char* min_env = getenv("VD_WSK_MIN_BLOCKS");
    const int min_blocks = min_env ? atoi(min_env) : 128;
    if (d.split_k > 1) return false;   // an explicit split factor asks for the split kernel
    if (nskip > 0 || nchunks < 4 || nchunks > WK_MAXC) return false;
    if ((d.ldc & 7) || ((d.flags & VD_EPI_RESIDUAL) && (d.ldr & 7))) return false;
    if ((d.flags & VD_EPI_RESIDUAL) && (d.flags & VD_EPI_ROWVEC) && d.act != VD_ACT_NONE) return false;
    return (d.M / 128) * (d.N / 32) >= min_blocks;
}
// split factor of the split kernel for `tiles` output tiles and `nchunks` chunks (the launcher's rule)
int wstream_split(const VdGemmDesc& d, int tiles, int nchunks, int* cps_out) {
    int var = g_ws_variant.load(std::memory_order_relaxed), target = g_ws_blocks.load(std::memory_order_relaxed);
    if (var < 0) {   // defaults (vd_conv3x3_wstream_set_variant changes them: development hook)
        var = 0;
        target = 256;
        g_ws_variant.store(var, std::memory_order_relaxed);
        g_ws_blocks.store(target, std::memory_order_relaxed);
    }
    int nsplit = d.split_k > 0 ? d.split_k : (target + tiles / 2) / tiles;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > nchunks) nsplit = nchunks;
    if (nsplit > VD_MAX_SPLIT_K) nsplit = VD_MAX_SPLIT_K;
    const int cps = (nchunks + nsplit - 1) / nsplit;
    if (cps_out) *cps_out = cps;
    return (nchunks + cps - 1) / cps;
}
}  // namespace

// Split factor vd_conv3x3_wstream_f16 will use for `desc` (size the workspace with it: VdGemmDesc.split_k); 0 = the whole-K
// kernel takes the launch and no workspace is needed.
extern "C" int vd_conv3x3_wstream_plan(const VdGemmDesc* dp, int* nsplit) {
    VD_REQUIRE(dp != nullptr && nsplit != nullptr, "vd_conv3x3_wstream_plan: null argument");
    VdGemmDesc d = *dp;
    if (d.c0 <= 0) d.c0 = d.K / 9;
    if (d.a1 == nullptr) d.c1 = 0;
    if (d.ldc <= 0) d.ldc = d.N;
    if (d.ldr <= 0) d.ldr = d.N;
    const int nchunks = (d.c0 + d.c1) / 64;
    const int nskip = (dp->skip_a0 != nullptr) ? (dp->skip_c0 + (dp->skip_a1 ? dp->skip_c1 : 0)) / 64 : 0;
    if (wsk_takes(d, nchunks, nskip)) { *nsplit = 0; return VD_OK; }
    const int ipb = wstream_ipb(d);
    *nsplit = wstream_split(d, (d.M / (64 * ipb)) * (d.N / (ipb == 4 ? 128 : 256)), nchunks, nullptr);
    return VD_OK;
}

extern "C" int vd_conv3x3_wstream_f16(const VdGemmDesc* dp, const void* w_stream, hipStream_t stream) {
    VD_REQUIRE(dp != nullptr && w_stream != nullptr, "vd_conv3x3_wstream_f16: null argument");
    VD_REQUIRE(dp->stat_sums == nullptr, "vd_conv3x3_wstream_f16: stat_sums is taken by vd_gemm_f16 / vd_ff_chain_f16 only");
    VdGemmDesc tmp = *dp;
    tmp.w = w_stream;          // the K-contiguous weights are not read on this path
    tmp.out_stats = nullptr;   // (validated below, not by the planner of the other kernels)
    // folded skip convolution: here skip_w holds the 1x1 weights in FRAGMENT order ([N / 32][chunks][4][64 lanes] x 16 bytes,
    // pack_linear_weight_stream); the planner of vd_gemm_f16 must not see the request (it only knows the halo kernel's form)
    const void* skip_a0 = dp->skip_a0; const void* skip_a1 = dp->skip_a1; const void* skip_w = dp->skip_w;
    tmp.skip_a0 = tmp.skip_a1 = tmp.skip_w = nullptr;
    GemmArgs a;
    const int rc = vd_gemm_normalise(&tmp, &a);
    if (rc != VD_OK) return rc;
    VdGemmDesc& d = a.d;
    const char* why = wstream_reject(d);
    VD_REQUIRE(why == nullptr, "vd_conv3x3_wstream_f16 takes %s", why ? why : "");
    VD_REQUIRE(((size_t)w_stream & 15) == 0, "vd_conv3x3_wstream_f16: w_stream must be 16-byte aligned");
    d.out_stats = dp->out_stats;
    d.sync = nullptr;
    if (d.out_stats != nullptr)
        VD_REQUIRE(d.N % 8 == 0 && d.ldc % 8 == 0 && (!(d.flags & VD_EPI_RESIDUAL) || d.ldr % 8 == 0) && ((size_t)d.out_stats & 7) == 0,
                   "vd_conv3x3_wstream_f16: out_stats needs 16-byte row segments");
    a.stat_rows = d.out_stats ? 64 : 0;
    WsArgs w;
    w.a0 = reinterpret_cast<const f16*>(d.a0); w.a1 = reinterpret_cast<const f16*>(d.a1);
    w.wp = reinterpret_cast<const uint4*>(w_stream); w.ws = d.ws;
    w.c0 = d.c0; w.c1 = d.c1; w.lda0 = d.lda0; w.lda1 = d.lda1;
    w.nimg = d.M / 64; w.M = d.M; w.N = d.N;
    w.nchunks = (d.c0 + d.c1) / 64;
    const int ipb = wstream_ipb(d);
    w.tiles_m = w.nimg / ipb; w.tiles_n = d.N / (ipb == 4 ? 128 : 256);
    w.a0_bytes = a.a0_bytes; w.a1_bytes = a.a1_bytes;
    w.s0 = nullptr; w.s1 = nullptr; w.swp = nullptr;
    w.sc0 = w.sc1 = w.slda0 = w.slda1 = w.nskip = w.skip_cps = 0;
    w.s0_bytes = w.s1_bytes = 0;
    if (skip_a0 != nullptr || skip_w != nullptr) {
        VD_REQUIRE(skip_a0 && skip_w && dp->skip_c0 > 0 && dp->skip_c0 % 64 == 0 && dp->skip_c1 % 64 == 0 && (skip_a1 || dp->skip_c1 == 0),
                   "vd_conv3x3_wstream_f16: folded skip convolution needs skip_a0, skip_w (fragment order) and channels in multiples of 64");
        VD_REQUIRE(((size_t)skip_w & 15) == 0, "vd_conv3x3_wstream_f16: skip_w must be 16-byte aligned");
        w.s0 = reinterpret_cast<const f16*>(skip_a0); w.s1 = reinterpret_cast<const f16*>(skip_a1);
        w.swp = reinterpret_cast<const uint4*>(skip_w);
        w.sc0 = dp->skip_c0; w.sc1 = skip_a1 ? dp->skip_c1 : 0;
        w.slda0 = dp->skip_lda0 > 0 ? dp->skip_lda0 : w.sc0;
        w.slda1 = dp->skip_lda1 > 0 ? dp->skip_lda1 : w.sc1;
        w.nskip = (w.sc0 + w.sc1) / 64;
        w.s0_bytes = (unsigned)((size_t)d.M * w.slda0 * 2);
        w.s1_bytes = (unsigned)((size_t)d.M * w.slda1 * 2);
    }
    if (wsk_takes(d, w.nchunks, w.nskip)) {   // whole K per block, epilogue in the kernel: no slabs, no reduce launch
        WkArgs k;
        k.w = w;
        k.w.tiles_m = w.nimg / WS_IPB;   // (the whole-K kernel works on image pairs)
        k.w.cps = w.nchunks;
        k.w.nsplit = 1;
        k.w.skip_cps = 0;
        k.g = a;
        {
            k.rotate = 1;
        }
        static std::atomic<unsigned long long> done{0};
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(done.load(std::memory_order_acquire) & bit)) {
            const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wsk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, WK_LDS);
            if (e != hipSuccess) {
                vd_set_error("vd_conv3x3_wstream_f16: cannot reserve %d bytes of LDS: %s", WK_LDS, hipGetErrorString(e));
                return VD_ERR_LAUNCH;
            }
            done.fetch_or(bit, std::memory_order_release);
        }
        hipLaunchKernelGGL(conv3x3_wsk_kernel, dim3(k.w.tiles_m * (d.N / 32)), dim3(WK_NT), WK_LDS, stream, k);
        return vd_check_launch("vd_conv3x3_wstream_f16/whole-K");
    }
    VD_REQUIRE(d.ws != nullptr, "vd_conv3x3_wstream_f16: needs the split-K workspace (vd_gemm_workspace_bytes)");
    // split over chunks until about one block per CU: 20 tiles -> 10 splits of 2 (4) chunks at the bench shape
    const int tiles = w.tiles_m * w.tiles_n;
    int nsplit = wstream_split(d, tiles, w.nchunks, &w.cps);
    const int var = g_ws_variant.load(std::memory_order_relaxed);
    w.nsplit = nsplit;
    w.skip_cps = (w.nskip + nsplit - 1) / nsplit;
    int lrc;
    switch (var) {
        case 1: lrc = launch_wstream<9, 1>(w, tiles * nsplit, stream); break;
        case 2: lrc = launch_wstream<4, 2>(w, tiles * nsplit, stream); break;
        case 3: lrc = launch_wstream<6, 2>(w, tiles * nsplit, stream); break;
        default: lrc = launch_wstream<12, 1>(w, tiles * nsplit, stream); break;
    }
    if (lrc != VD_OK) return lrc;
    return vd_gemm_launch_reduce(&a, nsplit, stream);
}


// ---- weight-streaming GEMM for the long-K, small-M projections (gemm_wstream_kernel.h) ----------------------------------
namespace {
const char* gw_reject(const VdGemmDesc& d) {
    if (d.ksize > 1 || d.stride > 1 || d.pad != 0 || d.ups != 0 || d.batch != 1 || d.a1 != nullptr) return "a plain single-source, unbatched GEMM";
    if (d.M % 128 != 0 || d.N % 256 != 0 || d.K % 64 != 0) return "M % 128 == 0, N % 256 == 0, K % 64 == 0";
    if (d.K / 64 > VD_MAX_SPLIT_K * GW_MAXC) return "K <= 65536 (the unrolled chunk sequence times the largest split)";
    if ((d.flags & (VD_EPI_LNFOLD | VD_EPI_OUT_F32 | VD_EPI_BIAS_ALONG_M)) || d.act == VD_ACT_GEGLU) return "a plain fp16 epilogue";
    return nullptr;
}
}  // namespace

extern "C" int vd_gemm_wstream_supported(const VdGemmDesc* dp) {
    if (dp == nullptr) return 0;
    VdGemmDesc d = *dp;
    if (d.batch <= 0) d.batch = 1;
    return gw_reject(d) == nullptr ? 1 : 0;
}

namespace {
int gw_split(const VdGemmDesc& d, int tiles, int nchunks) {
    const int target = 256;
    int nsplit = d.split_k > 0 ? d.split_k : (target + tiles / 2) / tiles;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > nchunks) nsplit = nchunks;
    if (nsplit > VD_MAX_SPLIT_K) nsplit = VD_MAX_SPLIT_K;
    if ((nchunks + nsplit - 1) / nsplit > GW_MAXC) nsplit = (nchunks + GW_MAXC - 1) / GW_MAXC;   // the kernel unrolls its chunks
    return nsplit;
}
}  // namespace

// Split factor vd_gemm_wstream_f16 will use for `desc` (size the workspace with it: VdGemmDesc.split_k).
extern "C" int vd_gemm_wstream_plan(const VdGemmDesc* dp, int* nsplit) {
    VD_REQUIRE(dp != nullptr && nsplit != nullptr, "vd_gemm_wstream_plan: null argument");
    const int nchunks = dp->K / 64;
    int ns = gw_split(*dp, (dp->M / 128) * (dp->N / 256), nchunks);
    const int cps = (nchunks + ns - 1) / ns;
    *nsplit = (nchunks + cps - 1) / cps;
    return VD_OK;
}

extern "C" int vd_gemm_wstream_f16(const VdGemmDesc* dp, const void* w_stream, hipStream_t stream) {
    VD_REQUIRE(dp != nullptr && w_stream != nullptr, "vd_gemm_wstream_f16: null argument");
    VD_REQUIRE(dp->stat_sums == nullptr, "vd_gemm_wstream_f16: stat_sums is taken by vd_gemm_f16 / vd_ff_chain_f16 only");
    VdGemmDesc tmp = *dp;
    tmp.w = w_stream;          // the K-contiguous weights are not read on this path
    tmp.out_stats = nullptr;   // (validated below, not by the planner of the other kernels)
    GemmArgs a;
    const int rc = vd_gemm_normalise(&tmp, &a);
    if (rc != VD_OK) return rc;
    VdGemmDesc& d = a.d;
    const char* why = gw_reject(d);
    VD_REQUIRE(why == nullptr, "vd_gemm_wstream_f16 takes %s", why ? why : "");
    VD_REQUIRE(d.ws != nullptr, "vd_gemm_wstream_f16: needs the split-K workspace (vd_gemm_workspace_bytes)");
    VD_REQUIRE(((size_t)w_stream & 15) == 0 && d.lda0 % 8 == 0, "vd_gemm_wstream_f16: w_stream must be 16-byte aligned, lda a multiple of 8");
    d.out_stats = dp->out_stats;
    d.sync = nullptr;
    if (d.out_stats != nullptr)
        VD_REQUIRE(d.N % 8 == 0 && d.ldc % 8 == 0 && (!(d.flags & VD_EPI_RESIDUAL) || d.ldr % 8 == 0) && ((size_t)d.out_stats & 7) == 0 && d.M % 64 == 0,
                   "vd_gemm_wstream_f16: out_stats needs 16-byte row segments");
    a.stat_rows = d.out_stats ? 64 : 0;
    GwArgs w;
    w.a = reinterpret_cast<const f16*>(d.a0); w.wp = reinterpret_cast<const uint4*>(w_stream); w.ws = d.ws;
    w.lda = d.lda0; w.M = d.M; w.N = d.N;
    w.nchunks = d.K / 64;
    w.tiles_m = d.M / 128; w.tiles_n = d.N / 256;
    w.a_bytes = a.a0_bytes;
    // split over the chunks until about one block per CU (a block = four waves, one per SIMD)
    const int tiles = w.tiles_m * w.tiles_n;
    int nsplit = gw_split(d, tiles, w.nchunks);
    VD_REQUIRE(nsplit <= VD_MAX_SPLIT_K, "vd_gemm_wstream_f16: K = %d needs more than %d splits", d.K, VD_MAX_SPLIT_K);
    w.cps = (w.nchunks + nsplit - 1) / nsplit;
    nsplit = (w.nchunks + w.cps - 1) / w.cps;
    w.nsplit = nsplit;
    hipLaunchKernelGGL(gemm_wstream_kernel, dim3(tiles * nsplit), dim3(256), 2 * GW_TILE, stream, w);
    const int lrc = vd_check_launch("vd_gemm_wstream_f16");
    if (lrc != VD_OK) return lrc;
    return vd_gemm_launch_reduce(&a, nsplit, stream);
}
