// Development probe (not part of libvd_hip.so): how many bytes per second the L2 -> LDS DMA path (buffer_load ... lds)
// delivers when every CU streams at once.  mode 0: every block walks the SAME `region` bytes (the weight-tile pattern of
// the fused kernels: one L2 line is wanted by 32 CUs of an XCD at about the same time); mode 1: every block its own region.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16(i32x4 rsrc, unsigned lds_base, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
template <int P>
__global__ __launch_bounds__(512, 2) void dma_probe_kernel(const char* src, unsigned region, int pieces, int mode, int skew) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const char* base = src + (mode == 1 ? (size_t)blockIdx.x * region : 0);
    const unsigned long long a = (unsigned long long)base;
    i32x4 rs;
    rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    rs.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    rs.z = (int)region;
    rs.w = 0x00020000;
    const unsigned voff = (unsigned)(wave * 1024 + lane * 16);
    unsigned soff = (unsigned)(((blockIdx.x * skew) * 8192) % region);
    for (int i = 0; i < pieces; ++i) {
        dma16(rs, lds0 + (unsigned)((wave * P + (i % P)) * 1024), voff, soff);
        soff += 8192;
        if (soff >= region) soff -= region;
        if (i >= P - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P - 1) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
extern "C" int dma_probe(const void* src, unsigned region, int blocks, int pieces, int mode, int depth, int skew, hipStream_t s) {
    if (depth == 4) hipLaunchKernelGGL(dma_probe_kernel<4>, dim3(blocks), dim3(512), 8 * 4 * 1024, s, (const char*)src, region, pieces, mode, skew);
    else if (depth == 8) hipLaunchKernelGGL(dma_probe_kernel<8>, dim3(blocks), dim3(512), 8 * 8 * 1024, s, (const char*)src, region, pieces, mode, skew);
    else hipLaunchKernelGGL(dma_probe_kernel<2>, dim3(blocks), dim3(512), 8 * 2 * 1024, s, (const char*)src, region, pieces, mode, skew);
    return (int)hipGetLastError();
}
