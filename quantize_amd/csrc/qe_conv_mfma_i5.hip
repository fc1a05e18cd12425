// qe_conv_mfma_i5.hip -- instantiations of the warp-specialised 3x3 MFMA kernel.
#include "qe_conv_mfma_kernel.hpp"

namespace qe {

#define QE_WS_K(NIW, SPLIT, NOPAD, RQ) \
    hipLaunchKernelGGL((conv_mfma_ws_kernel<NIW, 9, SPLIT, NOPAD, RQ>), dim3(blocks), dim3(2 * MF_THREADS), lds, s, a)
#define QE_WS(NIW, SPLIT)                                                                                        \
    do {                                                                                                          \
        const bool nopad = a.PADW == 0 && a.pad > 0;                                                              \
        if (a.rq_out != nullptr) { if (nopad) QE_WS_K(NIW, SPLIT, true, true); else QE_WS_K(NIW, SPLIT, false, true); }    \
        else { if (nopad) QE_WS_K(NIW, SPLIT, true, false); else QE_WS_K(NIW, SPLIT, false, false); }                       \
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
