// qe_conv_mfma_i5.hip -- instantiations of the warp-specialised 3x3 MFMA kernel.
#include "qe_conv_mfma_kernel.hpp"

namespace qe {

#define QE_WS(NIW, SPLIT)                                                                                        \
    do {                                                                                                          \
        if (a.PADW == 0 && a.pad > 0)                                                                             \
            hipLaunchKernelGGL((conv_mfma_ws_kernel<NIW, 9, SPLIT, true>), dim3(blocks), dim3(2 * MF_THREADS), lds, s, a);  \
        else                                                                                                      \
            hipLaunchKernelGGL((conv_mfma_ws_kernel<NIW, 9, SPLIT, false>), dim3(blocks), dim3(2 * MF_THREADS), lds, s, a); \
    } while (0)
#define QE_WS_SPLIT(NIW) \
    do { if (split == 4) QE_WS(NIW, 4); else if (split == 2) QE_WS(NIW, 2); else QE_WS(NIW, 1); } while (0)

void launch_mfma_ws(const MfmaArgs &a, int niw, int split, unsigned blocks, size_t lds, hipStream_t s)
{
    switch (niw) {
        case 7: QE_WS_SPLIT(7); break;
        case 4: QE_WS_SPLIT(4); break;
        default: QE_WS_SPLIT(2); break;
    }
}

}  // namespace qe
