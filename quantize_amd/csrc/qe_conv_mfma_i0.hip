// qe_conv_mfma_i0.hip -- instantiations of conv_mfma_kernel for the 4x1 wave layout.
#include "qe_conv_mfma_kernel.hpp"

namespace qe {

void launch_mfma_cfg0(const MfmaArgs &a, int niw, int ns, int KK, bool x8, unsigned blocks, size_t lds, hipStream_t s)
{
    switch (niw) {
        case 7: QE_MFMA_LAUNCH(4, 1, 7); break;
        case 4: QE_MFMA_LAUNCH(4, 1, 4); break;
        default: QE_MFMA_LAUNCH(4, 1, 2); break;
    }
}

}  // namespace qe
