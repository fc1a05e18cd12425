// qe_conv_mfma_i4.hip -- instantiations of the flat 1x1 MFMA kernel.
#include "qe_conv_mfma_kernel.hpp"

namespace qe {

#define QE_FLAT(WM, WN, NIW, NS, WRAW) \
    hipLaunchKernelGGL((conv_mfma_flat_kernel<WM, WN, NIW, NS, WRAW, false>), dim3(blocks), dim3(MF_THREADS), lds, s, a)

#define QE_FLAT_NS(WM, WN, NIW)                                     \
    do {                                                            \
        if (wraw) {                                                 \
            if (ns == 4) QE_FLAT(WM, WN, NIW, 4, true);             \
            else if (ns == 2) QE_FLAT(WM, WN, NIW, 2, true);        \
            else QE_FLAT(WM, WN, NIW, 1, true);                     \
        } else {                                                    \
            if (ns == 4) QE_FLAT(WM, WN, NIW, 4, false);            \
            else if (ns == 2) QE_FLAT(WM, WN, NIW, 2, false);       \
            else QE_FLAT(WM, WN, NIW, 1, false);                    \
        }                                                           \
    } while (0)

void launch_mfma_flat(const MfmaArgs &a, int cfg, int niw, int ns, bool wraw, bool s2, unsigned blocks, size_t lds, hipStream_t s)
{
    if (s2) {   // stride-2 1x1: 224-pixel tiles, 64-channel stages, 128-channel workgroups only
        if (wraw) hipLaunchKernelGGL((conv_mfma_flat_kernel<4, 1, 7, 2, true, true>), dim3(blocks), dim3(MF_THREADS), lds, s, a);
        else      hipLaunchKernelGGL((conv_mfma_flat_kernel<4, 1, 7, 2, false, true>), dim3(blocks), dim3(MF_THREADS), lds, s, a);
        return;
    }
    switch (cfg) {
        case 0:
            if (niw == 4) QE_FLAT_NS(4, 1, 4); else if (niw == 5) QE_FLAT_NS(4, 1, 5); else QE_FLAT_NS(4, 1, 7);
            break;
        case 1: QE_FLAT_NS(2, 2, 4); break;
        case 3:   // 8 waves x 32 output channels, 128-channel stages, packed weights (launch_conv_mfma: wide8)
        case 4:   // ... K loop unrolled over IC / 128 = 4 | 8 stages, activations two stages ahead
#define QE_FLAT8(NIW, NST) hipLaunchKernelGGL((conv_mfma_flat_kernel<8, 1, NIW, 4, true, false, false, NST>), dim3(blocks), dim3(512), lds, s, a)
            if (cfg == 4 && a.IC == 512) { if (niw == 5) QE_FLAT8(5, 4); else QE_FLAT8(7, 4); }
            else if (cfg == 4 && a.IC == 1024) { if (niw == 5) QE_FLAT8(5, 8); else QE_FLAT8(7, 8); }
            else { if (niw == 5) QE_FLAT8(5, 0); else QE_FLAT8(7, 0); }
#undef QE_FLAT8
            break;
        default: QE_FLAT_NS(1, 4, 2); break;
    }
}

}  // namespace qe
