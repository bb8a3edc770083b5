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

// ---- mask -> per-token weights of the masked CLIP image encoder (reference lib/model_zoo/clip.py:104-122) -------------------
// masks [B,1,H,W] -> clamp to [0,1] -> F.interpolate(bilinear, align_corners=False) to size x size -> per-patch mean
// (the reference's conv2d with a ones kernel / patch^2) and the global mean in front: out [B][1 + (size/patch)^2] fp32.
// One block per sample, one thread per patch (256 patches of 14x14 pixels for ViT-L/14 at 224): every interpolated
// pixel is 4 reads of an L2-resident mask; the global mean is the mean of the (equal-sized) patch means.
template <int KIND>
__device__ __forceinline__ float load_mask(const void* m, size_t i) {
    float v = KIND == 0 ? reinterpret_cast<const float*>(m)[i] : (float)reinterpret_cast<const f16*>(m)[i];
    return fminf(fmaxf(v, 0.f), 1.f);
}
template <int KIND>
__global__ __launch_bounds__(256) void mask_patch_weights_kernel(const void* masks, float* out, int H, int W, int size, int patch) {
    __shared__ float red[256];
    const int b = blockIdx.x, grid = size / patch, np = grid * grid;
    const float sy = (float)H / (float)size, sx = (float)W / (float)size;
    const size_t base = (size_t)b * H * W;
    float gsum = 0.f;
    for (int pidx = threadIdx.x; pidx < np; pidx += 256) {
        const int py = pidx / grid, px = pidx - py * grid;
        float acc = 0.f;
        for (int dy = 0; dy < patch; ++dy) {
            // torch's area_pixel_compute_source_index (align_corners = False): src = scale * (dst + 0.5) - 0.5, clamped at 0
            float fy = sy * ((float)(py * patch + dy) + 0.5f) - 0.5f;
            fy = fy < 0.f ? 0.f : fy;
            const int y0 = (int)fy;
            const int y1 = y0 + (y0 < H - 1 ? 1 : 0);
            const float ly = fy - (float)y0;
            for (int dx = 0; dx < patch; ++dx) {
                float fx = sx * ((float)(px * patch + dx) + 0.5f) - 0.5f;
                fx = fx < 0.f ? 0.f : fx;
                const int x0 = (int)fx;
                const int x1 = x0 + (x0 < W - 1 ? 1 : 0);
                const float lx = fx - (float)x0;
                const float v00 = load_mask<KIND>(masks, base + (size_t)y0 * W + x0), v01 = load_mask<KIND>(masks, base + (size_t)y0 * W + x1);
                const float v10 = load_mask<KIND>(masks, base + (size_t)y1 * W + x0), v11 = load_mask<KIND>(masks, base + (size_t)y1 * W + x1);
                acc += (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
            }
        }
        const float mean = acc / (float)(patch * patch);
        out[(size_t)b * (np + 1) + 1 + pidx] = mean;
        gsum += mean;
    }
    red[threadIdx.x] = gsum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[(size_t)b * (np + 1)] = red[0] / (float)np;
}

// ---- 'Simple' colour adjustment of image variation (reference app.py:373-379) ------------------------------------------------
// out[b][c] = clamp((img[b][c] - mean(img[b][c])) / std(img[b][c]) * std(ref[c]) + mean(ref[c]), 0, 1), std unbiased over
// the H*W pixels of a channel.  One block per (channel, image): statistics of both planes in fp32 (two sums each), then
// the affine map; planes of 512^2 halfs are L2-resident for the second read.
__global__ __launch_bounds__(256) void color_adjust_kernel(const f16* img, const f16* ref, f16* out, int HW, long ref_stride) {
    __shared__ float red[4][256];
    const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const f16* ip = img + ((size_t)b * 3 + c) * HW;
    const f16* rp = ref + (size_t)b * ref_stride + (size_t)c * HW;
    float s[4] = {0.f, 0.f, 0.f, 0.f};  // sum img, sumsq img (shifted), sum ref, sumsq ref (shifted)
    const float ki = (float)ip[0], kr = (float)rp[0];  // shift by the first sample: keeps the one-pass variance well conditioned
    for (int i = tid; i < HW; i += 256) {
        const float a = (float)ip[i] - ki, r = (float)rp[i] - kr;
        s[0] += a; s[1] += a * a; s[2] += r; s[3] += r * r;
    }
    for (int q = 0; q < 4; ++q) red[q][tid] = s[q];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o)
            for (int q = 0; q < 4; ++q) red[q][tid] += red[q][tid + o];
        __syncthreads();
    }
    const float n = (float)HW;
    const float mi = red[0][0] / n, mr = red[2][0] / n;
    const float vi = fmaxf((red[1][0] - n * mi * mi) / (n - 1.f), 0.f), vr = fmaxf((red[3][0] - n * mr * mr) / (n - 1.f), 0.f);
    // a constant output channel (std = 0) has nothing to stretch: it becomes the reference mean (the reference's 0 / 0 is NaN)
    const float scale = vi > 0.f ? sqrtf(vr) / sqrtf(vi) : 0.f, mean_i = mi + ki, mean_r = mr + kr;
    f16* op = out + ((size_t)b * 3 + c) * HW;
    for (int i = tid; i < HW; i += 256) {
        const float v = ((float)ip[i] - mean_i) * scale + mean_r;
        op[i] = (f16)fminf(fmaxf(v, 0.f), 1.f);
    }
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

extern "C" int vd_mask_patch_weights(const void* masks, int mask_kind, int B, int H, int W, int size, int patch, float* out,
                                     hipStream_t stream) {
    VD_REQUIRE(masks && out, "vd_mask_patch_weights: null pointer");
    VD_REQUIRE(B > 0 && H > 0 && W > 0, "vd_mask_patch_weights: empty input");
    VD_REQUIRE(mask_kind == 0 || mask_kind == 1, "vd_mask_patch_weights: mask_kind must be 0 (f32) or 1 (f16)");
    VD_REQUIRE(size > 0 && patch > 0 && size % patch == 0, "vd_mask_patch_weights: size=%d must be a multiple of patch=%d", size, patch);
    if (mask_kind == 0) hipLaunchKernelGGL(mask_patch_weights_kernel<0>, dim3(B), dim3(256), 0, stream, masks, out, H, W, size, patch);
    else hipLaunchKernelGGL(mask_patch_weights_kernel<1>, dim3(B), dim3(256), 0, stream, masks, out, H, W, size, patch);
    return vd_check_launch("vd_mask_patch_weights");
}

extern "C" int vd_color_adjust_f16(const void* img, const void* ref, void* out, int B, int H, int W, int64_t ref_batch_stride,
                                   hipStream_t stream) {
    VD_REQUIRE(img && ref && out, "vd_color_adjust_f16: null pointer");
    VD_REQUIRE(B > 0 && H > 0 && W > 0 && (size_t)H * W > 1, "vd_color_adjust_f16: empty input");
    hipLaunchKernelGGL(color_adjust_kernel, dim3(3, B), dim3(256), 0, stream, (const f16*)img, (const f16*)ref, (f16*)out, H * W,
                       (long)ref_batch_stride);
    return vd_check_launch("vd_color_adjust_f16");
}
