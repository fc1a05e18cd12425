// qe_conv_mfma_i3.hip -- instantiations of the small-IC (stem) MFMA kernel.
#include "qe_conv_mfma_kernel.hpp"

namespace qe {

void launch_mfma_smallic(const MfmaArgs &a, int cfg, int niw, unsigned blocks, size_t lds, hipStream_t s)
{
    if (cfg == 1 && niw == 7) {   // 64 output channels, 448-pixel tiles (4 rows of the 112-wide stem output)
        hipLaunchKernelGGL((conv_mfma_smallic_kernel<2, 2, 7>), dim3(blocks), dim3(MF_THREADS), lds, s, a);
        return;
    }
    switch (cfg) {
        case 0: hipLaunchKernelGGL((conv_mfma_smallic_kernel<4, 1, 7>), dim3(blocks), dim3(MF_THREADS), lds, s, a); break;
        case 1: hipLaunchKernelGGL((conv_mfma_smallic_kernel<2, 2, 4>), dim3(blocks), dim3(MF_THREADS), lds, s, a); break;
        default: hipLaunchKernelGGL((conv_mfma_smallic_kernel<1, 4, 2>), dim3(blocks), dim3(MF_THREADS), lds, s, a); break;
    }
}

}  // namespace qe
