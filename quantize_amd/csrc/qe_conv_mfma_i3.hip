// qe_conv_mfma_i3.hip -- instantiations of the small-IC (stem) MFMA kernel.
#include "qe_conv_mfma_kernel.hpp"

namespace qe {

#define QE_SMALLIC(WM, WN, NIW)                                                                                          \
    do {                                                                                                                \
        if (a.rq_out != nullptr && a.rq_patch && (NIW) == 7)                                                            \
            hipLaunchKernelGGL((conv_mfma_smallic_kernel<WM, WN, NIW, true, (NIW) == 7>), dim3(blocks), dim3(MF_THREADS), lds, s, a);  \
        else if (a.rq_out != nullptr)                                                                                   \
            hipLaunchKernelGGL((conv_mfma_smallic_kernel<WM, WN, NIW, true>), dim3(blocks), dim3(MF_THREADS), lds, s, a);  \
        else                                                                                                            \
            hipLaunchKernelGGL((conv_mfma_smallic_kernel<WM, WN, NIW, false>), dim3(blocks), dim3(MF_THREADS), lds, s, a); \
    } while (0)

void launch_mfma_smallic(const MfmaArgs &a, int cfg, int niw, unsigned blocks, size_t lds, hipStream_t s)
{
    if (cfg == 1 && niw == 7) {   // 64 output channels, 448-pixel tiles (4 rows of the 112-wide stem output)
        QE_SMALLIC(2, 2, 7);
        return;
    }
    switch (cfg) {
        case 0: QE_SMALLIC(4, 1, 7); break;
        case 1: QE_SMALLIC(2, 2, 4); break;
        default: QE_SMALLIC(1, 4, 2); break;
    }
}

}  // namespace qe
