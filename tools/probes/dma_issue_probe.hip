// LDS-DMA issue-cost probe (round 6): shader cycles one wave spends per `buffer_load_dwordx4 ... lds` (1 KiB) when it issues them
// back to back, next to MFMAs, and against the register path (global_load_dwordx4 + ds_write_b128), with 1 / 2 / 4 waves per SIMD.
// The source is a 64 KiB region per block (L2 / L1 resident after the first sweep).
//   hipcc --offload-arch=gfx950 -O3 -o dma_issue_probe tools/probes/dma_issue_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(i32x4 rsrc, unsigned lds_base, unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(lds_base), "v"(voff), "s"(rsrc) : "memory");
}

// MODE 0: NP DMA pieces per iteration, wait at the end of the iteration; 1: the same + 8 MFMAs interleaved; 2: 8 MFMAs only;
// 3: NP x (global_load_dwordx4 -> ds_write_b128) ; 4: the same + 8 MFMAs; 5: NP plain global_load_dwordx4 (no LDS)
template <int MODE, int NP>
__global__ __launch_bounds__(1024) void probe(const char* src, float* out, long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = src + (size_t)blockIdx.x * 65536;
    i32x4 rs;
    const unsigned long long a = (unsigned long long)base;
    rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    rs.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    rs.z = 65536; rs.w = 0x00020000;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + wave * (NP * 1024);
    f32x16 acc[4];
    f16x8 x, y;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int i = 0; i < 8; ++i) { x[i] = (f16)(0.01f * (lane & 7)); y[i] = (f16)0.5f; }
    uint4 keep = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const unsigned voff0 = (unsigned)(((it & 7) * 8192 + wave * 1024 + lane * 16) & 65535);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if constexpr (MODE == 1 || MODE == 2 || MODE == 4)
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[j & 3]) : "v"(x), "v"(y));
            if (j < NP) {
                const unsigned voff = (voff0 + j * 1024 * 13) & 65535;
                if constexpr (MODE == 0 || MODE == 1) dma16(rs, lds0 + j * 1024, voff);
                if constexpr (MODE == 3 || MODE == 4) {
                    const uint4 v = *reinterpret_cast<const uint4*>(base + voff);
                    *reinterpret_cast<uint4*>(smem + wave * (NP * 1024) + j * 1024 + lane * 16) = v;
                }
                if constexpr (MODE == 5) {
                    const uint4 v = *reinterpret_cast<const uint4*>(base + voff);
                    keep.x ^= v.x; keep.y ^= v.y; keep.z ^= v.z; keep.w ^= v.w;
                }
            }
        }
        if constexpr (MODE == 0 || MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    float s = keep.x + keep.y + keep.z + keep.w + smem[tid * 4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + tid] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}

template <int MODE, int NP>
void run(const char* what, int threads, const char* src) {
    const int blocks = 256, iters = 1000;
    float* out; long long* cyc;
    (void)hipMalloc(&out, sizeof(float) * threads * blocks);
    (void)hipMalloc(&cyc, sizeof(long long) * blocks * (threads / 64));
    const int lds = (threads / 64) * NP * 1024 + 4096;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MODE, NP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((probe<MODE, NP>), dim3(blocks), dim3(threads), lds, 0, src, out, cyc, iters);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, NP>), dim3(blocks), dim3(threads), lds, 0, src, out, cyc, iters);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const int nw = blocks * (threads / 64);
    long long* h = (long long*)malloc(sizeof(long long) * nw);
    (void)hipMemcpy(h, cyc, sizeof(long long) * nw, hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < nw; ++i) mean += (double)h[i]; mean /= nw;
    const double ns = ms * 1e6 / iters;
    const double gbs = (double)NP * 1024 * (threads / 64) / ns;   // per CU
    printf("%-46s NP %d waves/SIMD %d: %7.1f cyc/iter/wave %7.1f ns/iter  %6.1f GB/s per CU\n", what, NP, threads / 256, mean / iters, ns, (MODE == 2) ? 0.0 : gbs);
    (void)hipFree(out); (void)hipFree(cyc); free(h);
}

int main() {
    char* src; (void)hipMalloc(&src, 256 * 65536); (void)hipMemset(src, 1, 256 * 65536);
    for (int threads : {256, 512, 1024}) {
        run<2, 1>("8 MFMA only", threads, src);
        run<0, 1>("LDS-DMA x1 + wait", threads, src);
        run<0, 4>("LDS-DMA x4 + wait", threads, src);
        run<0, 8>("LDS-DMA x8 + wait", threads, src);
        run<1, 1>("8 MFMA + LDS-DMA x1", threads, src);
        run<1, 2>("8 MFMA + LDS-DMA x2", threads, src);
        run<1, 4>("8 MFMA + LDS-DMA x4", threads, src);
        run<1, 8>("8 MFMA + LDS-DMA x8", threads, src);
        run<3, 4>("(global_load -> ds_write) x4", threads, src);
        run<4, 2>("8 MFMA + (global_load -> ds_write) x2", threads, src);
        run<4, 4>("8 MFMA + (global_load -> ds_write) x4", threads, src);
        run<4, 8>("8 MFMA + (global_load -> ds_write) x8", threads, src);
        run<5, 8>("global_load_dwordx4 x8 (no LDS)", threads, src);
    }
    return 0;
}
