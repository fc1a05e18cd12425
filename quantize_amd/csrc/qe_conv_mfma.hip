// qe_conv_mfma.hip -- int8 MFMA implicit-GEMM convolution (placeholder until the kernel lands).
#include "qe_common.h"
namespace qe {
bool mfma_conv_eligible(const qe_conv_shape *, const qe_qparam *, const qe_qparam *) { return false; }
size_t mfma_conv_workspace_bytes(const qe_conv_shape *) { return 0; }
int launch_conv_mfma(const qe_qparam *, const qe_qparam *, const float *, const qe_conv_shape *, float *, void *,
                     size_t, hipStream_t) { return QE_ERR_UNSUPPORTED; }
}  // namespace qe
