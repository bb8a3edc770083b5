// Shared device/host helpers for the vd_hip kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 f16;
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define VD_OK 0
#define VD_ERR_ARG (-1)
#define VD_ERR_LAUNCH (-2)
#define VD_ERR_UNSUPPORTED (-3)

// Thread-local last-error text, see vd_last_error() in capi.hip
void vd_set_error(const char* fmt, ...);
int vd_check_launch(const char* what);

#define VD_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            vd_set_error(__VA_ARGS__);        \
            return VD_ERR_ARG;                \
        }                                     \
    } while (0)

union U4H8 {
    uint4 u;
    f16x8 h;
    f16 e[8];
};
union U2H4 {
    uint2 u;
    f16x4 h;
    f16 e[4];
};

__device__ __forceinline__ float vd_silu(float x) { return x / (1.0f + __expf(-x)); }
// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, far below fp16 output resolution): one v_rcp + one v_exp
// instead of ocml's branchy erff (~100 instructions), which dominated the GEGLU GEMM epilogue.
__device__ __forceinline__ float vd_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.44269504088896340736f);
    const float r = 1.0f - p * t * e;
    return copysignf(r, x);
}
// erf-form GELU (reference: F.gelu default, lib/model_zoo/attention.py:44)
__device__ __forceinline__ float vd_gelu_erf(float x) { return 0.5f * x * (1.0f + vd_erf(x * 0.70710678118654752440f)); }
// tanh-form GELU of the GPT-2 MLP (reference lib/model_zoo/optimus_models/optimus_gpt2.py:99-100); tanh(u) = 1 - 2 / (e^{2u} + 1)
__device__ __forceinline__ float vd_gelu_tanh(float x) {
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    const float t = 1.0f - 2.0f / (1.0f + __expf(2.0f * u));
    return 0.5f * x * (1.0f + t);
}
// quick GELU used by the HF CLIP towers (x * sigmoid(1.702 x))
__device__ __forceinline__ float vd_quick_gelu(float x) { return x / (1.0f + __expf(-1.702f * x)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- buffer descriptors and LDS-DMA (global memory -> LDS without a VGPR round trip) ----
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 make_rsrc_words(const void* p, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));  // stride 0, no swizzle
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}
// 16 bytes per lane straight from global memory into LDS (no VGPR round trip, no ds_write): LDS address =
// lds_base (wave-uniform, via M0) + lane * 16; the global side keeps the per-lane offset, so the XOR swizzle of the
// LDS image is applied by permuting WHICH 16-byte slot each lane fetches.  Issued through inline asm on purpose:
// hipcc would otherwise put s_waitcnt vmcnt(0) in front of every ds_read that follows (it cannot prove the DMA
// targets another stage), serialising load and MFMA phases.  Completion is tracked by hand (vmcnt) in the K loop.
__device__ __forceinline__ void dma16(i32x4 rsrc, unsigned lds_base, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :
                 : "s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt range");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

