// qe_conv_mfma_i7.hip -- instantiations of the flat 1x1 kernel for small planes (several whole images per tile).
#include "qe_conv_mfma_kernel.hpp"

namespace qe {

void launch_mfma_flatg(const MfmaArgs &a, int ns, bool wraw, unsigned blocks, size_t lds, hipStream_t s)
{
#define QE_FG(NS, WRAW) hipLaunchKernelGGL((conv_mfma_flatg_kernel<7, NS, WRAW, 7>), dim3(blocks), dim3(MF_THREADS), lds, s, a)   // 49-pixel planes
    if (wraw) {
        if (ns == 4) QE_FG(4, true); else QE_FG(2, true);
    } else {
        if (ns == 4) QE_FG(4, false); else QE_FG(2, false);
    }
#undef QE_FG
}

}  // namespace qe
