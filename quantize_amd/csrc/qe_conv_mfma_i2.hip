// qe_conv_mfma_i2.hip -- instantiations of conv_mfma_kernel for the 1x4 wave layout.
#include "qe_conv_mfma_kernel.hpp"

namespace qe {

void launch_mfma_cfg2(const MfmaArgs &a, int niw, int ns, int KK, bool x8, unsigned blocks, size_t lds, hipStream_t s)
{
    switch (niw) {
        case 2: QE_MFMA_LAUNCH(1, 4, 2); break;
        default: QE_MFMA_LAUNCH(1, 4, 1); break;
    }
}

}  // namespace qe
