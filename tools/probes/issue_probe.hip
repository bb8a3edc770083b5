// Issue-rate probe (round 6): how many shader cycles does ONE wave need for k MFMAs, for n transcendental / plain VALU
// instructions, and for both interleaved in one instruction stream?  One wave per SIMD (256-thread blocks, 1 block per CU) or
// two (512 threads); cycles from s_memtime around 2000 iterations of the unrolled body.
//   hipcc --offload-arch=gfx950 -O3 -o issue_probe tools/probes/issue_probe.hip && ./issue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512) void probe(float* out, long long* cyc, int iters) {
    f32x16 a0, a1, a2;
    f16x8 x, y;
    for (int i = 0; i < 16; ++i) { a0[i] = threadIdx.x * 1e-3f; a1[i] = i; a2[i] = -i; }
    for (int i = 0; i < 8; ++i) { x[i] = (f16)(threadIdx.x & 7); y[i] = (f16)0.001f; }
    float e[16];
    for (int i = 0; i < 16; ++i) e[i] = -0.001f * (threadIdx.x + i);
    float v[24];
    for (int i = 0; i < 24; ++i) v[i] = 0.5f + i;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0 || MODE == 2 || MODE == 4 || MODE == 6 || MODE == 7) {
            // 7 MFMAs: chain of 3 on a0, then a1 / a2 alternating (the attention step)
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y));
            if constexpr (MODE == 2) { for (int j = 0; j < 2; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(e[j])); }
            if constexpr (MODE == 4) { for (int j = 0; j < 3; ++j) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[j])); }
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y));
            if constexpr (MODE == 2) { for (int j = 2; j < 4; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(e[j])); }
            if constexpr (MODE == 4) { for (int j = 3; j < 6; ++j) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[j])); }
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(a0) : "v"(x), "v"(y));
            if constexpr (MODE == 2) { for (int j = 4; j < 7; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(e[j])); }
            if constexpr (MODE == 4) { for (int j = 6; j < 10; ++j) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[j])); }
            if constexpr (MODE == 6) { for (int j = 0; j < 4; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(e[j])); }
            if constexpr (MODE == 7) { for (int j = 0; j < 2; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(e[j])); }
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(a1) : "v"(x), "v"(y));
            if constexpr (MODE == 2) { for (int j = 7; j < 10; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(e[j])); }
            if constexpr (MODE == 4) { for (int j = 10; j < 14; ++j) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[j])); }
            if constexpr (MODE == 6) { for (int j = 4; j < 8; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(e[j])); }
            if constexpr (MODE == 7) { for (int j = 2; j < 4; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(e[j])); }
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(a2) : "v"(x), "v"(y));
            if constexpr (MODE == 2) { for (int j = 10; j < 12; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(e[j])); }
            if constexpr (MODE == 4) { for (int j = 14; j < 18; ++j) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[j])); }
            if constexpr (MODE == 6) { for (int j = 8; j < 12; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(e[j])); }
            if constexpr (MODE == 7) { for (int j = 4; j < 6; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(e[j])); }
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(a1) : "v"(x), "v"(y));
            if constexpr (MODE == 2) { for (int j = 12; j < 14; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(e[j])); }
            if constexpr (MODE == 4) { for (int j = 18; j < 21; ++j) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[j])); }
            if constexpr (MODE == 6) { for (int j = 12; j < 16; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(e[j])); }
            if constexpr (MODE == 7) { for (int j = 6; j < 8; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(e[j])); }
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(a2) : "v"(x), "v"(y));
            if constexpr (MODE == 2) { for (int j = 14; j < 16; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(e[j])); }
            if constexpr (MODE == 4) { for (int j = 21; j < 24; ++j) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[j])); }
        }
        if constexpr (MODE == 1) { for (int j = 0; j < 16; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(e[j])); }
        if constexpr (MODE == 3) { for (int j = 0; j < 24; ++j) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[j])); }
        if constexpr (MODE == 5) { for (int j = 0; j < 8; ++j) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[j]) : "v"(v[j + 8])); }
        if constexpr (MODE == 8) { for (int j = 0; j < 8; ++j) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[0]) : "v"(v[j + 8]), "v"(v[j + 9])); }
        if constexpr (MODE == 9) { for (int j = 0; j < 16; ++j) asm volatile("v_exp_f16 %0, %0" : "+v"(e[j])); }
        if constexpr (MODE == 10 || MODE == 11) {
            // role split between the two waves of a SIMD (512-thread blocks: waves w and w + 4 share one): one issues only the 7
            // MFMAs, the other only VALU (MODE 10: 16 v_exp_f32, MODE 11: 24 v_fma_f32)
            if (threadIdx.x < 256) {
                for (int j = 0; j < 7; ++j) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(j < 3 ? a0 : (j & 1 ? a1 : a2)) : "v"(x), "v"(y));
            } else if constexpr (MODE == 10) {
                for (int j = 0; j < 16; ++j) asm volatile("v_exp_f32 %0, %0" : "+v"(e[j]));
            } else {
                for (int j = 0; j < 24; ++j) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[j]));
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + e[i];
    for (int i = 0; i < 24; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
void run(const char* what, int threads, int blocks) {
    float* out; long long* cyc;
    hipMalloc(&out, sizeof(float) * threads * blocks);
    hipMalloc(&cyc, sizeof(long long) * blocks * (threads / 64));
    const int iters = 2000;
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int nw = blocks * (threads / 64);
    long long* h = (long long*)malloc(sizeof(long long) * nw);
    hipMemcpy(h, cyc, sizeof(long long) * nw, hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < nw; ++i) mean += (double)h[i]; mean /= nw;
    // s_memtime ticks at 100 MHz on gfx9 (constant clock), so report wall ns per iteration from the event as well
    printf("%-44s threads %4d blocks %4d: %8.1f memtime-ticks/iter  %8.1f ns/iter (wall)\n", what, threads, blocks, mean / iters, ms * 1e6 / iters);
    hipFree(out); hipFree(cyc); free(h);
}

int main() {
    for (int rep = 0; rep < 2; ++rep)
    for (int threads : {256, 512}) {
        const int blocks = 256;
        run<0>("7 MFMA 32x32x16 (3-chain + 2x2)", threads, blocks);
        run<1>("16 v_exp_f32", threads, blocks);
        run<2>("7 MFMA + 16 v_exp_f32 interleaved", threads, blocks);
        run<6>("3-chain, then 4 x (MFMA + 4 exp)", threads, blocks);
        run<7>("3-chain, then 4 x (MFMA + 2 exp)", threads, blocks);
        run<3>("24 v_fma_f32", threads, blocks);
        run<4>("7 MFMA + 24 v_fma_f32 interleaved", threads, blocks);
        run<5>("8 v_cvt_pk_f16_f32", threads, blocks);
        run<8>("8 v_max3_f32 (dependent chain)", threads, blocks);
        run<9>("16 v_exp_f16", threads, blocks);
        if (threads == 512) {
            run<10>("waves 0-3: 7 MFMA | waves 4-7: 16 v_exp_f32", threads, blocks);
            run<11>("waves 0-3: 7 MFMA | waves 4-7: 24 v_fma_f32", threads, blocks);
        }
    }
    return 0;
}
