// qe_conv_mfma_i6.hip -- instantiations of the two-strips-per-wave 3x3 MFMA kernel.
#include "qe_conv_mfma_kernel.hpp"

namespace qe {

template <int WMS, int SPLIT, bool RQ, bool PATCH = false>
static void launch_rq(const MfmaArgs &a, unsigned blocks, size_t lds, hipStream_t s)
{
    // more than 64 KB of dynamic LDS needs the attribute once per kernel
    static const bool raised = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma_sm2_kernel<WMS, 9, SPLIT, RQ, PATCH>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, MF_MAX_LDS_SM2) == hipSuccess;
    (void)raised;
    hipLaunchKernelGGL((conv_mfma_sm2_kernel<WMS, 9, SPLIT, RQ, PATCH>), dim3(blocks), dim3(MF_THREADS), lds, s, a);
}
template <int WMS, int SPLIT>
static void launch_one(const MfmaArgs &a, unsigned blocks, size_t lds, hipStream_t s)
{
    if (a.rq_out != nullptr && a.rq_patch) launch_rq<WMS, SPLIT, true, true>(a, blocks, lds, s);
    else if (a.rq_out != nullptr) launch_rq<WMS, SPLIT, true>(a, blocks, lds, s);
    else launch_rq<WMS, SPLIT, false>(a, blocks, lds, s);
}

void launch_mfma_sm2(const MfmaArgs &a, int wms, int split, unsigned blocks, size_t lds, hipStream_t s)
{
    if (wms == 2) {
        if (split == 4) launch_one<2, 4>(a, blocks, lds, s);
        else if (split == 2) launch_one<2, 2>(a, blocks, lds, s);
        else launch_one<2, 1>(a, blocks, lds, s);
    } else {
        if (split == 4) launch_one<1, 4>(a, blocks, lds, s);
        else if (split == 2) launch_one<1, 2>(a, blocks, lds, s);
        else launch_one<1, 1>(a, blocks, lds, s);
    }
}

}  // namespace qe
