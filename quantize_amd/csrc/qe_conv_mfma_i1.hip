// qe_conv_mfma_i1.hip -- instantiations of conv_mfma_kernel for the 2x2 wave layout.
#include "qe_conv_mfma_kernel.hpp"

namespace qe {

void launch_mfma_cfg1(const MfmaArgs &a, int niw, int ns, int KK, bool x8, unsigned blocks, size_t lds, hipStream_t s)
{
    switch (niw) {
        case 4: QE_MFMA_LAUNCH(2, 2, 4); break;
        case 2: QE_MFMA_LAUNCH(2, 2, 2); break;
        default: QE_MFMA_LAUNCH(2, 2, 1); break;
    }
}

}  // namespace qe
