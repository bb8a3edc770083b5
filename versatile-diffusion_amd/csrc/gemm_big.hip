// One-wave-per-SIMD instances of the GEMM kernel template: 4 waves per block, one block per CU, per-wave tiles of
// 64x160 / 64x128 / 128x64 (10 / 8 MFMA tiles).  A wave may use the whole 512-entry register file of its SIMD, so the
// accumulators (128-160 registers), two sets of operand fragments and the epilogue prefetch fit without spilling, and the
// LDS read traffic per MFMA drops from 1.0-1.2 KiB (64x64 / 32x160 wave tiles) to 0.7-0.75 KiB -- LDS bandwidth (operand
// reads + the DMA writes) is what bounds the two-waves-per-SIMD instances.  Built WITHOUT -amdgpu-mfma-vgpr-form: the
// accumulators live in AGPRs, leaving the 256 architectural VGPRs to fragments, addresses and the epilogue.
#include "gemm_kernel.h"

int vd_gemm_launch_big(int cfg, int variant, const void* args, int nsplit, hipStream_t stream) {
    const GemmArgs& a = *reinterpret_cast<const GemmArgs*>(args);
    if (variant == 1) {   // LayerNorm-fold instantiation
        if (cfg == 24) return launch_cfg<256, 256, 64, 128, 512, 2, 64, 2, true>(a, nsplit, stream);
        vd_set_error("vd_gemm_f16: no LayerNorm-fold instance of tile configuration %d", cfg);
        return VD_ERR_ARG;
    }
    switch (cfg) {
        case 8: return launch_cfg<128, 320, 64, 160, 256, 2, 64, 1>(a, nsplit, stream);
        case 9: return launch_cfg<128, 256, 64, 128, 256, 3, 64, 1>(a, nsplit, stream);
        case 10: return launch_cfg<256, 128, 128, 64, 256, 3, 64, 1>(a, nsplit, stream);
        case 12: return launch_cfg<128, 320, 64, 160, 256, 4, 32, 1>(a, nsplit, stream);
        // 8 waves, two per SIMD, 64x160 / 64x128 wave tiles (single fragment set): the largest tiles the register file
        // and LDS admit -- half the L2->LDS and LDS->register traffic per FLOP of the 128x160 / 128x128 tiles
        case 22: return launch_cfg<256, 320, 64, 160, 512, 2, 64, 2>(a, nsplit, stream);
        case 23: return launch_cfg<256, 320, 64, 160, 512, 4, 32, 2>(a, nsplit, stream);
        case 24: return launch_cfg<256, 256, 64, 128, 512, 2, 64, 2>(a, nsplit, stream);
        default:
            vd_set_error("vd_gemm_f16: unknown tile configuration %d", cfg);
            return VD_ERR_ARG;
    }
}
