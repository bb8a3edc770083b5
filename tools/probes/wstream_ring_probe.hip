// Weight-stream ring probe (round 6): every block of a 256-block launch streams the same W bytes of weights through an LDS ring
// of NSLOT slots of TILE bytes by LDS-DMA (8 waves, each wave TILE / 8192 pieces per tile), wait + barrier per tile as the projection
// stages of ff_chain / rowchain do, with or without MFMA work per tile.  Prints us per tile for ring depths / tile sizes.
//   hipcc --offload-arch=gfx950 -O3 -o wstream_ring_probe tools/probes/wstream_ring_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(i32x4 rsrc, unsigned lds_base, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// TILE_KB: tile size in KiB (multiple of 8: every wave issues TILE_KB / 8 pieces), NSLOT ring slots, MF: MFMAs per tile and wave
template <int TILE_KB, int NSLOT, int MF, bool PRIVATE>
__global__ __launch_bounds__(512) void probe(const char* w, float* out, int ntiles) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr int PW = TILE_KB / 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = w + (PRIVATE ? (size_t)blockIdx.x * ((size_t)ntiles * TILE_KB * 1024) : 0);
    i32x4 rs;
    const unsigned long long a = (unsigned long long)base;
    rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    rs.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    rs.z = ntiles * TILE_KB * 1024; rs.w = 0x00020000;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    f32x16 acc[2];
    f16x8 x, y;
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int i = 0; i < 8; ++i) { x[i] = (f16)(0.01f * (lane & 7)); y[i] = (f16)0.5f; }
    auto issue = [&](int t) {
        const unsigned dst = lds0 + (unsigned)((t % NSLOT) * TILE_KB * 1024);
#pragma unroll
        for (int j = 0; j < PW; ++j) dma16(rs, dst + (unsigned)((j * 8 + wave) * 1024), (unsigned)(((j * 8 + wave) * 1024) + lane * 16), (unsigned)(t * TILE_KB * 1024));
    };
#pragma unroll
    for (int t = 0; t < NSLOT - 1; ++t) issue(t);
    for (int t = 0; t < ntiles; ++t) {
        if (t + NSLOT - 1 <= ntiles) wait_vm<PW*(NSLOT - 2)>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + NSLOT - 1 < ntiles) issue(t + NSLOT - 1);
        const f16x8 a0 = *reinterpret_cast<const f16x8*>(smem + (t % NSLOT) * TILE_KB * 1024 + lane * 16);
#pragma unroll
        for (int m = 0; m < MF; ++m) acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, (m & 2) ? x : y, acc[m & 1], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + tid] = s;
}

template <int TILE_KB, int NSLOT, int MF, bool PRIVATE>
void run(const char* w, float* out, int total_kb) {
    const int ntiles = total_kb / TILE_KB, blocks = 256, lds = TILE_KB * 1024 * NSLOT;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<TILE_KB, NSLOT, MF, PRIVATE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((probe<TILE_KB, NSLOT, MF, PRIVATE>), dim3(blocks), dim3(512), lds, 0, w, out, ntiles);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((probe<TILE_KB, NSLOT, MF, PRIVATE>), dim3(blocks), dim3(512), lds, 0, w, out, ntiles);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / 10;
    printf("tile %3d KiB x %d slots, %2d MFMA/tile, %s source, %5d KiB per block: %7.1f us per launch  %6.3f us per tile  %6.1f GB/s per CU\n", TILE_KB, NSLOT, MF,
           PRIVATE ? "private" : "shared ", total_kb, us, us / ntiles, total_kb * 1024.0 / (us * 1e3));
}

int main() {
    char* w; float* out;
    const int total_kb = 1600;   // per block (private: 256 x 1600 KiB = 400 MiB)
    (void)hipMalloc(&w, (size_t)256 * total_kb * 1024); (void)hipMemset(w, 1, (size_t)256 * total_kb * 1024);
    (void)hipMalloc(&out, 256 * 512 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        run<16, 3, 10, false>(w, out, total_kb);
        run<16, 4, 10, false>(w, out, total_kb);
        run<16, 6, 10, false>(w, out, total_kb);
        run<16, 9, 10, false>(w, out, total_kb);
        run<8, 4, 5, false>(w, out, total_kb);
        run<8, 8, 5, false>(w, out, total_kb);
        run<8, 16, 5, false>(w, out, total_kb);
        run<16, 3, 0, false>(w, out, total_kb);
        run<16, 9, 0, false>(w, out, total_kb);
        run<16, 3, 10, true>(w, out, total_kb);
        run<16, 9, 10, true>(w, out, total_kb);
        run<32, 4, 20, false>(w, out, total_kb);
    }
    return 0;
}
