// CLIP image pre-processing on the device, bit-exact with the host path of the reference
// (/root/reference/lib/model_zoo/clip.py:88-94: torchvision ToPILImage -> HuggingFace CLIPProcessor, i.e. Pillow's 8-bit
// bicubic resize of the shortest edge to 224, centre crop, rescale by 1/255 in double, normalise in float32).
//
// Byte / integer work, HBM-bound and tiny (a 512x512 image is 0.8 MB): two passes like Pillow's ImagingResample --
// horizontal into a uint8 scratch (only the columns the crop keeps), then vertical + crop + normalise.  The taps are
// Pillow's own fixed-point coefficients (22 fractional bits), computed on the host in double (vd_hip/resample.py) so the
// device only does integer multiply-adds, the rounding shift and the clip; the final float32 values come from a
// 256 x 3 table of (level * (1/255) - mean) / std evaluated on the host exactly as numpy does.
#include "vd_common.h"
#include "../../include/vd_hip.h"

namespace {

constexpr int RES_PRECISION_BITS = 22;

__device__ __forceinline__ int to_u8(float v) { return (int)(unsigned char)(v * 255.0f); }  // ToPILImage: mul(255).byte()
__device__ __forceinline__ int clip8(int acc) {
    const int v = acc >> RES_PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

template <int KIND>  // 0: float32, 1: float16, 2: uint8 source, all [B,3,H,W]
__device__ __forceinline__ int load_level(const void* img, size_t idx) {
    if constexpr (KIND == 0) return to_u8(reinterpret_cast<const float*>(img)[idx]);
    else if constexpr (KIND == 1) return (int)(unsigned char)(float)(reinterpret_cast<const f16*>(img)[idx] * (f16)255.0f);
    else return (int)reinterpret_cast<const uint8_t*>(img)[idx];
}

// tmp[b][c][y][x'] for x' in [0, size): horizontal pass at output column crop_l + x' (or a plain copy when rw == W)
template <int KIND>
__global__ __launch_bounds__(256) void resample_h_kernel(const void* img, uint8_t* tmp, int H, int W, int size, int crop_l,
                                                         const int32_t* hb, const int32_t* hk, int hks, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int xo = (int)(i % size);
    const size_t row = i / size;  // (b * 3 + c) * H + y
    const size_t src = row * (size_t)W;
    const int x = crop_l + xo;
    int lv;
    if (hks == 0) {
        lv = load_level<KIND>(img, src + x);
    } else {
        const int x0 = hb[2 * x], n = hb[2 * x + 1];
        int acc = 1 << (RES_PRECISION_BITS - 1);
        for (int t = 0; t < n; ++t) acc += load_level<KIND>(img, src + x0 + t) * hk[(size_t)x * hks + t];
        lv = clip8(acc);
    }
    tmp[i] = (uint8_t)lv;
}

// out[b][c][y'][x'] = table[level][c], level = vertical pass over tmp at output row crop_t + y'
__global__ __launch_bounds__(256) void resample_v_norm_kernel(const uint8_t* tmp, f16* out, int H, int size, int crop_t,
                                                              const int32_t* vb, const int32_t* vk, int vks,
                                                              const float* table, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int xo = (int)(i % size);
    const int yo = (int)((i / size) % size);
    const size_t bc = i / ((size_t)size * size);
    const int c = (int)(bc % 3);
    const uint8_t* col = tmp + bc * (size_t)H * size + xo;
    const int y = crop_t + yo;
    int lv;
    if (vks == 0) {
        lv = col[(size_t)y * size];
    } else {
        const int y0 = vb[2 * y], n = vb[2 * y + 1];
        int acc = 1 << (RES_PRECISION_BITS - 1);
        for (int t = 0; t < n; ++t) acc += (int)col[(size_t)(y0 + t) * size] * vk[(size_t)y * vks + t];
        lv = clip8(acc);
    }
    out[i] = (f16)table[lv * 3 + c];
}

// decoded image [B,3,H,W] (float32 / float16 in [0,1]) -> uint8 [B,H,W,3]: torchvision ToPILImage = mul(255).byte() in the
// tensor's own dtype, then CHW -> HWC (reference app.py:319 on the output of vae_decode)
template <int KIND>
__global__ __launch_bounds__(256) void image_to_u8_kernel(const void* img, uint8_t* out, int HW, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // output index: ((b * HW + p) * 3 + c)
    if (i >= total) return;
    const int c = (int)(i % 3);
    const size_t bp = i / 3;
    const size_t b = bp / HW, p = bp - b * HW;
    out[i] = (uint8_t)load_level<KIND>(img, (b * 3 + c) * (size_t)HW + p);
}

}  // namespace

extern "C" int vd_image_to_u8(const void* img, int img_kind, int B, int H, int W, uint8_t* out, hipStream_t stream) {
    VD_REQUIRE(img && out, "vd_image_to_u8: null pointer");
    VD_REQUIRE(B > 0 && H > 0 && W > 0, "vd_image_to_u8: empty input");
    VD_REQUIRE(img_kind == 0 || img_kind == 1, "vd_image_to_u8: img_kind must be 0 (f32) or 1 (f16)");
    const size_t total = (size_t)B * H * W * 3;
    const unsigned g = (unsigned)((total + 255) / 256);
    if (img_kind == 0) hipLaunchKernelGGL(image_to_u8_kernel<0>, dim3(g), dim3(256), 0, stream, img, out, H * W, total);
    else hipLaunchKernelGGL(image_to_u8_kernel<1>, dim3(g), dim3(256), 0, stream, img, out, H * W, total);
    return vd_check_launch("vd_image_to_u8");
}

extern "C" int vd_clip_preprocess_f16(const void* img, int img_kind, int B, int H, int W, int rh, int rw, const int32_t* hb,
                                      const int32_t* hk, int hks, const int32_t* vb, const int32_t* vk, int vks, int crop_t,
                                      int crop_l, int size, const float* norm_table, uint8_t* tmp, void* out,
                                      hipStream_t stream) {
    VD_REQUIRE(img && norm_table && tmp && out, "vd_clip_preprocess_f16: null pointer");
    VD_REQUIRE(B > 0 && H > 0 && W > 0 && size > 0, "vd_clip_preprocess_f16: empty input");
    VD_REQUIRE(img_kind >= 0 && img_kind <= 2, "vd_clip_preprocess_f16: img_kind must be 0 (f32), 1 (f16) or 2 (u8)");
    VD_REQUIRE(rh >= size && rw >= size && crop_t >= 0 && crop_l >= 0 && crop_t + size <= rh && crop_l + size <= rw,
               "vd_clip_preprocess_f16: crop %dx%d at (%d,%d) does not fit the resized image %dx%d", size, size, crop_t, crop_l, rh, rw);
    VD_REQUIRE((hks == 0) == (rw == W) && (vks == 0) == (rh == H),
               "vd_clip_preprocess_f16: a pass has taps iff its axis is resized (hks=%d rw=%d W=%d, vks=%d rh=%d H=%d)", hks, rw, W, vks, rh, H);
    VD_REQUIRE((hks == 0 || (hb && hk)) && (vks == 0 || (vb && vk)), "vd_clip_preprocess_f16: missing coefficient table");
    const size_t th = (size_t)B * 3 * H * size;
    const unsigned gh = (unsigned)((th + 255) / 256);
    if (img_kind == 0) hipLaunchKernelGGL(resample_h_kernel<0>, dim3(gh), dim3(256), 0, stream, img, tmp, H, W, size, crop_l, hb, hk, hks, th);
    else if (img_kind == 1) hipLaunchKernelGGL(resample_h_kernel<1>, dim3(gh), dim3(256), 0, stream, img, tmp, H, W, size, crop_l, hb, hk, hks, th);
    else hipLaunchKernelGGL(resample_h_kernel<2>, dim3(gh), dim3(256), 0, stream, img, tmp, H, W, size, crop_l, hb, hk, hks, th);
    const size_t tv = (size_t)B * 3 * size * size;
    hipLaunchKernelGGL(resample_v_norm_kernel, dim3((unsigned)((tv + 255) / 256)), dim3(256), 0, stream, tmp, (f16*)out, H, size,
                       crop_t, vb, vk, vks, norm_table, tv);
    return vd_check_launch("vd_clip_preprocess_f16");
}
