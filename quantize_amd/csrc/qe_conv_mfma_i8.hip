// qe_conv_mfma_i8.hip -- instantiations of the flat 1x1 MFMA kernel for 4-bit activations read from the packed stream
// (128-channel workgroups, prepared weight fragments).
#include "qe_conv_mfma_kernel.hpp"

namespace qe {

#define QE_FLAT4(NIW, NS) \
    hipLaunchKernelGGL((conv_mfma_flat_kernel<4, 1, NIW, NS, false, false, true>), dim3(blocks), dim3(MF_THREADS), lds, s, a)
#define QE_FLAT4_NS(NIW) \
    do { if (ns == 4) QE_FLAT4(NIW, 4); else if (ns == 2) QE_FLAT4(NIW, 2); else QE_FLAT4(NIW, 1); } while (0)

void launch_mfma_flat_x4(const MfmaArgs &a, int niw, int ns, unsigned blocks, size_t lds, hipStream_t s)
{
    if (niw == 4) QE_FLAT4_NS(4); else if (niw == 5) QE_FLAT4_NS(5); else QE_FLAT4_NS(7);
}

}  // namespace qe
