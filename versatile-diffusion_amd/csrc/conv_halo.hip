// Launch planner and the two-waves-per-SIMD instances of conv3x3_halo_kernel (conv_halo_kernel.h); the one-wave-per-SIMD
// instances live in conv_halo_big.hip.  Reached from vd_gemm_f16 (gemm.hip) for every 3x3 / stride 1 / pad 1 convolution
// whose geometry the halo formulation accepts.
#include "conv_halo_kernel.h"

namespace {

struct HaloVariant { int bm, bn, nt, mode; const char* name; };
// index = variant id (tile_cfg of vd_gemm_plan = vd_gemm_num_configs() + id)
const HaloVariant kHalo[VD_CONV_HALO_VARIANTS] = {
    {256, 160, 512, 0, "conv3x3_halo_kernel<256,160,32,160,512,0>"},
    {256, 160, 512, 1, "conv3x3_halo_kernel<256,160,32,160,512,1>"},
    {256, 160, 512, 2, "conv3x3_halo_kernel<256,160,32,160,512,2>"},
    {256, 128, 512, 0, "conv3x3_halo_kernel<256,128,64,64,512,0>"},
    {256, 128, 512, 1, "conv3x3_halo_kernel<256,128,64,64,512,1>"},
    {256, 128, 512, 2, "conv3x3_halo_kernel<256,128,64,64,512,2>"},
    {256, 160, 256, 1, "conv3x3_halo_kernel<256,160,64,160,256,1>"},
    {256, 160, 256, 2, "conv3x3_halo_kernel<256,160,64,160,256,2>"},
    {256, 160, 512, 3, "conv3x3_halo_kernel<256,160,32,160,512,3>"},
    {256, 128, 512, 3, "conv3x3_halo_kernel<256,128,64,64,512,3>"},
    {128, 32, 256, 4, "conv3x3_halo_kernel<128,32,32,32,256,4>"},
    {128, 160, 256, 2, "conv3x3_halo_kernel<128,160,32,160,256,2>"},
    {256, 160, 512, 2, "conv3x3_halo_kernel<256,160,32,160,512,2,skip>"},
};

std::atomic<int> g_halo_variant{-2};   // -2: not read from the environment yet; -1: planner; 0: off; k > 0: force variant k - 1

int halo_setting() {
    int v = g_halo_variant.load(std::memory_order_relaxed);
    if (v == -2) {
        const char* e = getenv("VD_CONV_HALO");
        v = e ? atoi(e) : -1;
        if (v < -1 || v > VD_CONV_HALO_VARIANTS || (v > 0 && !(v - 1 == 2 || v - 1 == 5 || v - 1 == 12))) v = -1;
        g_halo_variant.store(v, std::memory_order_relaxed);
    }
    return v;
}

inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// patch geometry for BM output pixels per block; false when the convolution does not fit the halo formulation
bool halo_geometry(const GemmArgs& a, int BM, ConvHaloArgs& c) {
    const VdGemmDesc& d = a.d;
    if (d.ksize != 3 || d.stride != 1 || d.pad != 1 || d.batch != 1) return false;
    if (d.flags & (VD_EPI_LNFOLD | VD_EPI_OUT_F32 | VD_EPI_BIAS_ALONG_M)) return false;
    if (d.act == VD_ACT_GEGLU) return false;
    if (d.c0 % 64 != 0 || d.c1 % 64 != 0 || d.N % 8 != 0) return false;
    const int Hv = d.Hin << d.ups, Wv = d.Win << d.ups;
    if (d.Hout != Hv || d.Wout != Wv) return false;   // symmetric zero padding only
    const long npix = (long)Hv * Wv;
    if (d.M % npix != 0) return false;
    const long nimg = d.M / npix;
    if (nimg * d.Hin * d.Win >= (1l << 28)) return false;   // packed (pixel << 3 | slot) source indices
    const int tw = (Wv % 32 == 0) ? 32 : (Wv % 16 == 0) ? 16 : (Wv % 8 == 0) ? 8 : 0;
    if (tw == 0) return false;
    const int th = BM / tw;
    int ngrp, rg;
    if (Hv % th == 0) {
        ngrp = 1;
        rg = th;
    } else if (th % Hv == 0 && tw == Wv && is_pow2(Hv) && nimg % (th / Hv) == 0) {
        ngrp = th / Hv;   // whole small images per patch
        rg = Hv;
    } else {
        return false;
    }
    c.g = a;
    c.tw = tw;
    c.ltw = ilog2(tw);
    c.rg = rg;
    c.ngrp = ngrp;
    c.lgsz = ilog2(tw * rg);
    c.pitch = tw + 2;
    c.gpx = (rg + 2) * c.pitch;
    c.hpx = ngrp * c.gpx;
    if (c.hpx > BM * 100 / 64 + 16) return false;
    c.mg_pitch = (1 << 20) / c.pitch + 1;
    c.mg_gpx = (1 << 20) / c.gpx + 1;
    c.tiles_x = Wv / tw;
    c.tiles_y = ngrp == 1 ? Hv / rg : 1;
    c.nchunks = (d.c0 + d.c1) / 64;
    c.chunks_per_split = c.nchunks;
    c.Hv = Hv;
    c.Wv = Wv;
    c.halo_bytes = ((c.hpx + 7) / 8) * 1024;
    c.g.tiles_m = (int)(d.M / BM);
    {
        static const char* abl_env = getenv("VD_HALO_ABL");
        c.abl = abl_env ? atoi(abl_env) : 0;
    }
    return true;
}

}  // namespace

// Instantiated: variants 2 (256 x 160 blocks, pinned mid-barrier loop: every UNet width), 5 (256 x 128: the VAE widths) and 12 (2 with
// the folded skip convolution).  The other table slots were development variants (barrier placements 0 / 1 / 3, one wave per SIMD,
// whole-K small-M blocks, 128-pixel patches) that lost their in-forward A/Bs (profiles/HISTORY.md); removed in round 6.
static bool halo_built(int variant) { return variant == 2 || variant == 5 || variant == 12; }

extern "C" int vd_conv_halo_set_variant(int v) {
    VD_REQUIRE(v >= -1 && v <= VD_CONV_HALO_VARIANTS, "vd_conv_halo_set_variant: %d out of range", v);
    VD_REQUIRE(v <= 0 || halo_built(v - 1), "vd_conv_halo_set_variant: variant %d is not instantiated", v - 1);
    g_halo_variant.store(v, std::memory_order_relaxed);
    return VD_OK;
}

const char* vd_conv_halo_name(int variant) {
    return (variant >= 0 && variant < VD_CONV_HALO_VARIANTS) ? kHalo[variant].name : nullptr;
}

// Decide whether (and how) the validated problem *gemm_args runs on the halo kernel.  Returns 1 and fills conv_args
// (a ConvHaloArgs), the variant id and the split factor, or 0 when the problem stays on gemm_f16_kernel.
int vd_conv_halo_plan(const void* gemm_args, int can_split, void* conv_args, int* variant_out, int* nsplit_out) {
    const GemmArgs& a = *static_cast<const GemmArgs*>(gemm_args);
    const VdGemmDesc& d = a.d;
    const int setting = halo_setting();
    if (setting == 0) return 0;
    int v;
    if (setting > 0) {
        v = setting - 1;
    } else {
        // the planner's choice, measured INSIDE the UNet forward (tools/shape_profile.py under VD_CONV_HALO=k,
        // tools/halo_forward.py): 256 x 160 blocks of 32 x 160 wave tiles with the pinned request order (MODE 2) win on every
        // UNet width (320 / 640 / 1280) -- although 64 x 64 wave tiles (256 x 128 blocks) are 5-15 % faster back to back with
        // warm weights (tools/halo_check.py time), they lose 0.2 ms per forward where weights stream cold; they serve the
        // widths 160 does not divide (the VAE's 128 / 256 / 512)
        if (d.N % 160 == 0) v = 2;
        else if (d.N % 128 == 0) v = 5;
        else return 0;
    }
    const HaloVariant& hv = kHalo[v];
    ConvHaloArgs& c = *static_cast<ConvHaloArgs*>(conv_args);
    if (!halo_geometry(a, hv.bm, c)) return 0;
    c.g.tiles_n = (d.N + hv.bn - 1) / hv.bn;
    const int wst = hv.mode == 0 ? 2 : 3;   // weight stages of the mode
    if (2 * c.halo_bytes + wst * (hv.mode == 4 ? 9 : 1) * hv.bn * 128 > 160 * 1024) return 0;
    const long tiles = (long)c.g.tiles_m * c.g.tiles_n;
    if (setting < 0 && tiles < 32) {
        // 8x8-level layers: too few 256-pixel patches; they go to the weight-streaming kernels (conv_wstream.hip) or stay on
        // gemm_f16_kernel with a deep split over K.  (A whole-K small-M instance of this kernel -- 128-pixel patches x 32 channels --
        // measured 0.09 ms per forward slower, round 4; removed.)
        return 0;
    }
    // split over channel chunks until one round of blocks covers the chip; unit = one tap of one block
    int ns = 1;
    if (d.split_k > 0) {
        ns = d.split_k;
    } else if (can_split) {
        float best = 1e30f;
        for (int s = 1; s <= c.nchunks && s <= VD_MAX_SPLIT_K / 2; ++s) {
            const int cps = (c.nchunks + s - 1) / s;
            if ((c.nchunks + cps - 1) / cps != s) continue;   // same work per block as a smaller factor
            const long rounds = (tiles * s + 255) / 256;
            const float t = (float)rounds * (cps * 9 + 8) + (s > 1 ? 6.f + 2.f * s : 0.f);
            if (t < best) { best = t; ns = s; }
        }
    }
    if (ns > c.nchunks) ns = c.nchunks;
    if (ns > 1 && !can_split) ns = 1;
    c.chunks_per_split = (c.nchunks + ns - 1) / ns;
    ns = (c.nchunks + c.chunks_per_split - 1) / c.chunks_per_split;
    c.nskip = 0;
    c.skip_cps = 0;
    if (d.skip_a0 != nullptr && d.skip_w != nullptr) {
        // folded 1x1 skip convolution: the planner's 256 x 160 instance only, inputs on the output grid
        if (v != 2 || d.ups != 0 || d.skip_c0 % 64 != 0 || d.skip_c1 % 64 != 0 || d.skip_c0 <= 0 || (d.skip_a1 == nullptr && d.skip_c1 != 0)) return 0;
        const size_t rows = (size_t)d.M;
        const size_t b0 = rows * (size_t)d.skip_lda0 * 2, b1 = d.skip_a1 ? rows * (size_t)d.skip_lda1 * 2 : 0, bw = (size_t)d.N * d.skip_ldw * 2;
        if (b0 >= (1ull << 31) || b1 >= (1ull << 31) || bw >= (1ull << 31)) return 0;
        c.nskip = (d.skip_c0 + d.skip_c1) / 64;
        c.skip_cps = (c.nskip + ns - 1) / ns;
        c.s0_bytes = (unsigned)b0;
        c.s1_bytes = (unsigned)b1;
        c.sw_bytes = (unsigned)bw;
        v = 12;
    }
    *variant_out = v;
    *nsplit_out = ns;
    return 1;
}


int vd_conv_halo_launch(const void* conv_args, int variant, int nsplit, hipStream_t stream) {
    const ConvHaloArgs& c = *static_cast<const ConvHaloArgs*>(conv_args);
    switch (variant) {
        case 2: return launch_conv_halo<256, 160, 32, 160, 512, 2>(c, nsplit, stream);
        case 5: return launch_conv_halo<256, 128, 64, 64, 512, 2>(c, nsplit, stream);
        case 12: return launch_conv_halo<256, 160, 32, 160, 512, 2, true>(c, nsplit, stream);
        default:
            vd_set_error("conv3x3_halo: variant %d is not instantiated", variant);
            return VD_ERR_UNSUPPORTED;
    }
}
