// One-wave-per-SIMD instances of conv3x3_halo_kernel: 4 waves with 64 x 160 wave tiles (160 accumulator registers per
// lane), built without -amdgpu-mfma-vgpr-form so that the accumulators may live in AGPRs (see gemm_big.hip).
#include "conv_halo_kernel.h"

int vd_conv_halo_launch_big(const void* conv_args, int variant, int nsplit, hipStream_t stream) {
    const ConvHaloArgs& c = *static_cast<const ConvHaloArgs*>(conv_args);
    switch (variant) {
        case 6: return launch_conv_halo<256, 160, 64, 160, 256, 1>(c, nsplit, stream);
        case 7: return launch_conv_halo<256, 160, 64, 160, 256, 2>(c, nsplit, stream);
        default:
            vd_set_error("conv3x3_halo: unknown variant %d", variant);
            return VD_ERR_ARG;
    }
}
