// vd_conv3x3_wstream_f16: launcher and instances of conv3x3_wstream_kernel (conv_wstream_kernel.h).  Built WITHOUT
// -amdgpu-mfma-vgpr-form: the 128 accumulator registers of a wave live in AGPRs, the architectural VGPRs hold the ring of
// weight fragments in flight, the pixel fragments and the addresses.
#include "conv_wstream_kernel.h"

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

template <int D, int OCC>
int launch_wstream(const WsArgs& w, int blocks, hipStream_t stream) {
    hipLaunchKernelGGL((conv3x3_wstream_kernel<D, OCC>), dim3(blocks), dim3(256), 2 * WS_HB, stream, w);
    return vd_check_launch("vd_conv3x3_wstream_f16");
}
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

extern "C" int vd_conv3x3_wstream_f16(const VdGemmDesc* dp, const void* w_stream, hipStream_t stream) {
    VD_REQUIRE(dp != nullptr && w_stream != nullptr, "vd_conv3x3_wstream_f16: null argument");
    VdGemmDesc tmp = *dp;
    tmp.w = w_stream;          // the K-contiguous weights are not read on this path
    tmp.out_stats = nullptr;   // (validated below, not by the planner of the other kernels)
    GemmArgs a;
    const int rc = vd_gemm_normalise(&tmp, &a);
    if (rc != VD_OK) return rc;
    VdGemmDesc& d = a.d;
    const char* why = wstream_reject(d);
    VD_REQUIRE(why == nullptr, "vd_conv3x3_wstream_f16 takes %s", why ? why : "");
    VD_REQUIRE(d.ws != nullptr, "vd_conv3x3_wstream_f16: needs the split-K workspace (vd_gemm_workspace_bytes)");
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
    w.tiles_m = w.nimg / 2; w.tiles_n = d.N / 256;
    w.a0_bytes = a.a0_bytes; w.a1_bytes = a.a1_bytes;
    // split over chunks until about one block per CU: 20 tiles -> 10 splits of 2 (4) chunks at the bench shape
    const int tiles = w.tiles_m * w.tiles_n;
    int var = g_ws_variant.load(std::memory_order_relaxed), target = g_ws_blocks.load(std::memory_order_relaxed);
    if (var < 0) {   // development switches (VD_WSTREAM_VAR / VD_WSTREAM_BLOCKS or vd_conv3x3_wstream_set_variant), read once
        const char* var_env = getenv("VD_WSTREAM_VAR");
        const char* tgt_env = getenv("VD_WSTREAM_BLOCKS");
        var = var_env ? atoi(var_env) : 0;
        target = tgt_env ? atoi(tgt_env) : 256;
        g_ws_variant.store(var, std::memory_order_relaxed);
        g_ws_blocks.store(target, std::memory_order_relaxed);
    }
    int nsplit = d.split_k > 0 ? d.split_k : (target + tiles / 2) / tiles;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > w.nchunks) nsplit = w.nchunks;
    if (nsplit > VD_MAX_SPLIT_K) nsplit = VD_MAX_SPLIT_K;
    w.cps = (w.nchunks + nsplit - 1) / nsplit;
    nsplit = (w.nchunks + w.cps - 1) / w.cps;
    w.nsplit = nsplit;
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
