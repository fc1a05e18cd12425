// qe_api.hip -- C-ABI entry points that are not tied to one kernel file:
// error strings, version, and the conv dispatch (generic fp32 vs int8 MFMA).
#include "qe_common.h"

#include <atomic>
#include <cstring>
#include <mutex>
#include <string>
#include <utility>
#include <vector>
#include <unistd.h>

extern char **environ;

namespace qe {

struct RequantHost {   // qe_conv_mfma_kernel.hpp
    uint8_t *out;
    const float *scale, *zero;
    int n_param;
    float qmin, qmax;
    int n_bits, sign;
    int32_t *status;
};

thread_local int g_last_hip_error = 0;

namespace {
struct EnvSnapshot {
    std::vector<std::pair<std::string, std::string>> kv;
};
std::atomic<EnvSnapshot *> g_env{nullptr};
std::mutex g_env_mutex;
EnvSnapshot *take_env_snapshot()
{
    auto *snap = new EnvSnapshot;
    for (char **e = environ; e != nullptr && *e != nullptr; ++e) {
        if (std::strncmp(*e, "QE_", 3) != 0) continue;
        const char *eq = std::strchr(*e, '=');
        if (eq == nullptr) continue;
        snap->kv.emplace_back(std::string(*e, eq - *e), std::string(eq + 1));
    }
    return snap;
}
}  // namespace

const char *env_get(const char *name)
{
    EnvSnapshot *snap = g_env.load(std::memory_order_acquire);
    if (snap == nullptr) {
        std::lock_guard<std::mutex> lock(g_env_mutex);
        snap = g_env.load(std::memory_order_acquire);
        if (snap == nullptr) {
            snap = take_env_snapshot();
            g_env.store(snap, std::memory_order_release);
        }
    }
    for (const auto &kv : snap->kv)
        if (kv.first == name) return kv.second.c_str();
    return nullptr;
}

int launch_conv_generic(bool packed_in, const void *x, const qe_qparam *xq, const qe_qparam *w,
                        const float *bias, const qe_conv_shape *sh, float *out, hipStream_t s);

// qe_conv_mfma.hip
bool mfma_conv_eligible(const qe_conv_shape *sh, const qe_qparam *x, const qe_qparam *w);
size_t mfma_conv_workspace_bytes(const qe_conv_shape *sh, int x_bits, int w_bits);
size_t mfma_conv_prepared_bytes(const qe_conv_shape *sh, int x_bits, int w_bits);
uint64_t mfma_conv_prepared_layout(const qe_conv_shape *sh, int x_bits, int w_bits);
int launch_conv_mfma(const qe_qparam *x, const qe_qparam *w, const float *bias, const qe_conv_shape *sh,
                     float *out, void *workspace, size_t workspace_bytes, hipStream_t s, int mode, void *prepared,
                     size_t prepared_bytes, const RequantHost *rq = nullptr);
bool mfma_conv_requant_fused(const qe_conv_shape *sh, const qe_qparam *x, const qe_qparam *w, int rq_bits, int rq_n_param);

// qe_conv_f32.hip
bool f32_conv_eligible(const qe_conv_shape *sh, const qe_qparam *w);
size_t f32_conv_prepared_bytes(const qe_conv_shape *sh);
int launch_conv_f32(const float *x, const qe_qparam *w, const float *bias, const qe_conv_shape *sh, float *out,
                    void *prepared, size_t prepared_bytes, hipStream_t s, int mode);

static int check_shape(const qe_conv_shape *sh)
{
    if (sh == nullptr) return QE_ERR_ARG;
    if (sh->N < 0 || sh->IC <= 0 || sh->H <= 0 || sh->W <= 0 || sh->OC < 0 || sh->KH <= 0 || sh->KW <= 0 ||
        sh->stride <= 0 || sh->padding < 0)
        return QE_ERR_ARG;
    return QE_OK;
}

// scale / zero arrays: one element, or one per channel (IC for activations, OC for weights) -- anything else would be
// indexed out of bounds by channel (the reference reads scale[ic] / scale[oc] unchecked, quantconv2d.cu:112-127)
static int check_nparam(const qe_qparam *x, const qe_qparam *w, const qe_conv_shape *sh)
{
    if (x != nullptr && !(x->n_param == 1 || x->n_param >= sh->IC)) return QE_ERR_ARG;
    if (!(w->n_param == 1 || w->n_param >= sh->OC)) return QE_ERR_ARG;
    return QE_OK;
}

static int check_qparam(const qe_qparam *q)
{
    if (q == nullptr || q->data == nullptr || q->scale == nullptr || q->zero == nullptr) return QE_ERR_ARG;
    if (!(q->n_bits > 0 && q->n_bits <= 8)) return QE_ERR_NBITS;
    if (q->n_param < 1) return QE_ERR_ARG;
    return QE_OK;
}

}  // namespace qe

extern "C" const char *qe_error_string(int status)
{
    switch (status) {
        case QE_OK: return "ok";
        case QE_ERR_NBITS: return "n_bits must be in the range (0, 8]";          // tpack.cu:13
        case QE_ERR_RANGE: return "The input tensor is out of range.";           // tpack.cu:14
        case QE_ERR_DTYPE: return "unsupported element type";
        case QE_ERR_ARG: return "invalid argument";
        case QE_ERR_HIP: return "HIP runtime error";
        case QE_ERR_WORKSPACE: return "workspace too small";
        case QE_ERR_UNSUPPORTED: return "problem shape not supported by the gfx950 kernels";
        default: return "unknown error";
    }
}

extern "C" int qe_last_hip_error(void) { return qe::g_last_hip_error; }

// Not part of the public ABI (absent from include/quant_engine.h): take a fresh snapshot of the QE_* environment knobs.
// Old snapshots are kept alive (a few hundred bytes each): a concurrent reader may still hold a pointer into one.
extern "C" void qe_debug_reload_env(void)
{
    std::lock_guard<std::mutex> lock(qe::g_env_mutex);
    qe::g_env.store(qe::take_env_snapshot(), std::memory_order_release);
}
extern "C" const char *qe_version(void) { return "quantize_amd 0.1.0"; }
extern "C" const char *qe_target_arch(void) { return "gfx950"; }

extern "C" size_t qe_quantconv2d_workspace_bytes(const qe_conv_shape *shape, int x_bits, int w_bits)
{
    if (qe::check_shape(shape) != QE_OK) return 0;
    return qe::mfma_conv_workspace_bytes(shape, x_bits, w_bits);
}

extern "C" int qe_quantconv2d_path(const qe_conv_shape *shape, const qe_qparam *x, const qe_qparam *w)
{
    if (qe::check_shape(shape) != QE_OK || x == nullptr || w == nullptr) return 0;
    return qe::mfma_conv_eligible(shape, x, w) ? 1 : 0;
}

extern "C" int qe_quantconv2d(const qe_qparam *x, const qe_qparam *w, const float *bias,
                              const qe_conv_shape *shape, float *out,
                              void *workspace, size_t workspace_bytes, qe_stream_t stream)
{
    using namespace qe;
    int rc = check_shape(shape);
    if (rc != QE_OK) return rc;
    if ((rc = check_qparam(x)) != QE_OK) return rc;
    if ((rc = check_qparam(w)) != QE_OK) return rc;
    if ((rc = check_nparam(x, w, shape)) != QE_OK) return rc;
    if (out == nullptr) return QE_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (mfma_conv_eligible(shape, x, w))
        return launch_conv_mfma(x, w, bias, shape, out, workspace, workspace_bytes, s, 0, nullptr, 0);
    return launch_conv_generic(true, x->data, x, w, bias, shape, out, s);
}

extern "C" size_t qe_conv_prepared_bytes(const qe_conv_shape *shape, int x_bits, int w_bits)
{
    if (qe::check_shape(shape) != QE_OK) return 0;
    return qe::mfma_conv_prepared_bytes(shape, x_bits, w_bits);
}

extern "C" uint64_t qe_conv_prepared_layout(const qe_conv_shape *shape, int x_bits, int w_bits)
{
    if (qe::check_shape(shape) != QE_OK) return 0;
    return qe::mfma_conv_prepared_layout(shape, x_bits, w_bits);
}

extern "C" size_t qe_quantconv2d_prepared_workspace_bytes(const qe_conv_shape *shape, int x_bits, int w_bits)
{
    if (qe::check_shape(shape) != QE_OK) return 0;
    return qe::mfma_conv_workspace_bytes(shape, x_bits, w_bits) - qe::mfma_conv_prepared_bytes(shape, x_bits, w_bits);
}

extern "C" int qe_conv_prepare(const qe_qparam *w, const float *bias, const qe_conv_shape *shape, int x_bits,
                               void *prepared, size_t prepared_bytes, qe_stream_t stream)
{
    using namespace qe;
    int rc = check_shape(shape);
    if (rc != QE_OK) return rc;
    if ((rc = check_qparam(w)) != QE_OK) return rc;
    if ((rc = check_nparam(nullptr, w, shape)) != QE_OK) return rc;
    if (!(x_bits > 0 && x_bits <= 8)) return QE_ERR_NBITS;
    if (mfma_conv_prepared_bytes(shape, x_bits, w->n_bits) == 0) return QE_OK;     // nothing to prepare for this problem
    qe_qparam x = *w;                      // only n_bits / n_param of the activations select the plan
    x.n_bits = x_bits; x.n_param = 1;
    return launch_conv_mfma(&x, w, bias, shape, nullptr, nullptr, 0, static_cast<hipStream_t>(stream), 1, prepared, prepared_bytes);
}

extern "C" int qe_quantconv2d_prepared(const qe_qparam *x, const qe_qparam *w, const float *bias,
                                       const qe_conv_shape *shape, const void *prepared, size_t prepared_bytes,
                                       float *out, void *workspace, size_t workspace_bytes, qe_stream_t stream)
{
    using namespace qe;
    int rc = check_shape(shape);
    if (rc != QE_OK) return rc;
    if ((rc = check_qparam(x)) != QE_OK) return rc;
    if ((rc = check_qparam(w)) != QE_OK) return rc;
    if ((rc = check_nparam(x, w, shape)) != QE_OK) return rc;
    if (out == nullptr) return QE_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (mfma_conv_eligible(shape, x, w))
        return launch_conv_mfma(x, w, bias, shape, out, workspace, workspace_bytes, s, 2, const_cast<void *>(prepared), prepared_bytes);
    return launch_conv_generic(true, x->data, x, w, bias, shape, out, s);   // per-channel activation scales: nothing is prepared
}

// ---- fused re-quantisation (conv-epilogue form) ----
static int check_requant(const qe_requant *rq)
{
    if (rq == nullptr || rq->scale == nullptr || rq->zero == nullptr) return QE_ERR_ARG;
    if (!(rq->n_bits > 0 && rq->n_bits <= 8)) return QE_ERR_NBITS;
    if (rq->n_param < 1) return QE_ERR_ARG;
    return QE_OK;
}
static size_t requant_y_bytes(const qe_conv_shape *sh)
{
    const int64_t OH = (sh->H + 2 * sh->padding - sh->KH) / sh->stride + 1, OW = (sh->W + 2 * sh->padding - sh->KW) / sh->stride + 1;
    if (OH <= 0 || OW <= 0) return 0;
    return ((size_t)sh->N * sh->OC * OH * OW * sizeof(float) + 255) / 256 * 256;
}

extern "C" int qe_quantconv2d_requant_path(const qe_conv_shape *shape, const qe_qparam *x, const qe_qparam *w, const qe_requant *rq)
{
    if (qe::check_shape(shape) != QE_OK || x == nullptr || w == nullptr || rq == nullptr) return 0;
    if (!qe::mfma_conv_eligible(shape, x, w)) return 0;
    return qe::mfma_conv_requant_fused(shape, x, w, rq->n_bits, rq->n_param) ? 1 : 0;
}

extern "C" size_t qe_quantconv2d_requant_workspace_bytes(const qe_conv_shape *shape, const qe_qparam *x, const qe_qparam *w,
                                                         const qe_requant *rq)
{
    if (qe::check_shape(shape) != QE_OK || x == nullptr || w == nullptr || rq == nullptr) return 0;
    const size_t conv = (qe_quantconv2d_prepared_workspace_bytes(shape, x->n_bits, w->n_bits) + 255) / 256 * 256;
    return qe_quantconv2d_requant_path(shape, x, w, rq) ? conv : conv + requant_y_bytes(shape);
}

extern "C" int qe_quantconv2d_requant_prepared(const qe_qparam *x, const qe_qparam *w, const float *bias,
                                               const qe_conv_shape *shape, const void *prepared, size_t prepared_bytes,
                                               const qe_requant *rq, uint8_t *out, int32_t *status,
                                               void *workspace, size_t workspace_bytes, qe_stream_t stream)
{
    using namespace qe;
    int rc = check_shape(shape);
    if (rc != QE_OK) return rc;
    if ((rc = check_qparam(x)) != QE_OK) return rc;
    if ((rc = check_qparam(w)) != QE_OK) return rc;
    if ((rc = check_nparam(x, w, shape)) != QE_OK) return rc;
    if ((rc = check_requant(rq)) != QE_OK) return rc;
    if (out == nullptr) return QE_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t OH = (shape->H + 2 * shape->padding - shape->KH) / shape->stride + 1;
    const int64_t OW = (shape->W + 2 * shape->padding - shape->KW) / shape->stride + 1;
    if (OH <= 0 || OW <= 0) return QE_ERR_ARG;
    if (shape->N == 0 || shape->OC == 0) return QE_OK;
    const size_t conv_ws = (qe_quantconv2d_prepared_workspace_bytes(shape, x->n_bits, w->n_bits) + 255) / 256 * 256;
    if (qe_quantconv2d_requant_path(shape, x, w, rq)) {
        const RequantHost rh{out, rq->scale, rq->zero, rq->n_param, rq->qmin, rq->qmax, rq->n_bits, rq->sign, status};
        rc = launch_conv_mfma(x, w, bias, shape, nullptr, workspace, workspace_bytes, s, 2, const_cast<void *>(prepared),
                              prepared_bytes, &rh);
        return rc;   // (a kernel without the epilogue here would contradict qe_quantconv2d_requant_path: surface it)
    }
    // two passes: y in fp32 behind the conv scratch, then the fused quantise + pack kernel (bit-identical by construction)
    const size_t ybytes = requant_y_bytes(shape);
    if (workspace == nullptr || workspace_bytes < conv_ws + ybytes) return QE_ERR_WORKSPACE;
    float *y = reinterpret_cast<float *>(static_cast<uint8_t *>(workspace) + conv_ws);
    rc = qe_quantconv2d_prepared(x, w, bias, shape, prepared, prepared_bytes, y, workspace, conv_ws, stream);
    if (rc != QE_OK) return rc;
    return qe_quantize_pack(y, (int64_t)shape->N * shape->OC * OH * OW, rq->scale, rq->zero, rq->n_param, OH * OW, rq->qmin,
                            rq->qmax, rq->n_bits, rq->sign, out, status, stream);
}

extern "C" int qe_quantconv2d_float_input(const float *x, const qe_qparam *w, const float *bias,
                                          const qe_conv_shape *shape, float *out, qe_stream_t stream)
{
    using namespace qe;
    int rc = check_shape(shape);
    if (rc != QE_OK) return rc;
    if ((rc = check_qparam(w)) != QE_OK) return rc;
    if ((rc = check_nparam(nullptr, w, shape)) != QE_OK) return rc;
    if (x == nullptr || out == nullptr) return QE_ERR_ARG;
    return launch_conv_generic(false, x, nullptr, w, bias, shape, out, static_cast<hipStream_t>(stream));
}

extern "C" int qe_quantconv2d_float_input_path(const qe_conv_shape *shape, const qe_qparam *w)
{
    if (qe::check_shape(shape) != QE_OK || w == nullptr) return 0;
    return qe::f32_conv_eligible(shape, w) ? 1 : 0;
}

extern "C" size_t qe_quantconv2d_float_input_workspace_bytes(const qe_conv_shape *shape, int w_bits)
{
    (void)w_bits;
    if (qe::check_shape(shape) != QE_OK) return 0;
    return qe::f32_conv_prepared_bytes(shape);
}

extern "C" int qe_quantconv2d_float_input_ws(const float *x, const qe_qparam *w, const float *bias,
                                             const qe_conv_shape *shape, float *out, void *workspace,
                                             size_t workspace_bytes, qe_stream_t stream)
{
    using namespace qe;
    int rc = check_shape(shape);
    if (rc != QE_OK) return rc;
    if ((rc = check_qparam(w)) != QE_OK) return rc;
    if ((rc = check_nparam(nullptr, w, shape)) != QE_OK) return rc;
    if (x == nullptr || out == nullptr) return QE_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (f32_conv_eligible(shape, w)) return launch_conv_f32(x, w, bias, shape, out, workspace, workspace_bytes, s, 0);
    return launch_conv_generic(false, x, nullptr, w, bias, shape, out, s);
}

extern "C" int qe_conv_f32_prepare(const qe_qparam *w, const float *bias, const qe_conv_shape *shape,
                                   void *prepared, size_t prepared_bytes, qe_stream_t stream)
{
    using namespace qe;
    int rc = check_shape(shape);
    if (rc != QE_OK) return rc;
    if ((rc = check_qparam(w)) != QE_OK) return rc;
    if ((rc = check_nparam(nullptr, w, shape)) != QE_OK) return rc;
    if (!f32_conv_eligible(shape, w)) return QE_OK;          // nothing to prepare: the VALU kernel reads the packed weights
    return launch_conv_f32(nullptr, w, bias, shape, nullptr, prepared, prepared_bytes, static_cast<hipStream_t>(stream), 1);
}

extern "C" int qe_quantconv2d_float_input_prepared(const float *x, const qe_qparam *w, const float *bias,
                                                   const qe_conv_shape *shape, const void *prepared,
                                                   size_t prepared_bytes, float *out, qe_stream_t stream)
{
    using namespace qe;
    int rc = check_shape(shape);
    if (rc != QE_OK) return rc;
    if ((rc = check_qparam(w)) != QE_OK) return rc;
    if ((rc = check_nparam(nullptr, w, shape)) != QE_OK) return rc;
    if (x == nullptr || out == nullptr) return QE_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (f32_conv_eligible(shape, w))
        return launch_conv_f32(x, w, bias, shape, out, const_cast<void *>(prepared), prepared_bytes, s, 2);
    return launch_conv_generic(false, x, nullptr, w, bias, shape, out, s);
}

// ---------------------------------------------------------------------------------------------
// Global average pool (bench.py's top-1 tail): a workgroup copies 64 planes (64 * P contiguous floats) into LDS with
// coalesced 16-byte loads, then 4 threads per plane add up a quarter each (P is small: 49 for ResNet-50) and the
// quarters are combined with two shuffles.  HBM-bound: 103 MB for (256, 2048, 7, 7).
// ---------------------------------------------------------------------------------------------
namespace qe {
constexpr int AP_PLANES = 64;
__global__ __launch_bounds__(256) void global_avgpool_kernel(const float *__restrict__ x, float *__restrict__ out,
                                                             int64_t n_planes, int P)
{
    extern __shared__ __attribute__((aligned(16))) float sp[];
    const int64_t plane0 = (int64_t)blockIdx.x * AP_PLANES;
    const int np = (int)((n_planes - plane0) < AP_PLANES ? (n_planes - plane0) : AP_PLANES);
    const int nf = np * P;
    const float *src = x + plane0 * P;
    const bool al = (reinterpret_cast<uintptr_t>(src) & 15) == 0;
    if (al) {
        for (int i = threadIdx.x * 4; i < nf; i += 256 * 4) {
            if (i + 4 <= nf) *reinterpret_cast<float4 *>(sp + i) = *reinterpret_cast<const float4 *>(src + i);
            else for (int j = i; j < nf; ++j) sp[j] = src[j];
        }
    } else {
        for (int i = threadIdx.x; i < nf; i += 256) sp[i] = src[i];
    }
    __syncthreads();
    const int pl = threadIdx.x >> 2, part = threadIdx.x & 3;
    float sum = 0.0f;
    if (pl < np) {
        const int per = (P + 3) >> 2;
        const int lo = part * per, hi = (lo + per < P) ? lo + per : P;
        for (int i = lo; i < hi; ++i) sum += sp[pl * P + i];
    }
    sum += __shfl_xor(sum, 1);
    sum += __shfl_xor(sum, 2);
    if (pl < np && part == 0) out[plane0 + pl] = sum / (float)P;
}
}  // namespace qe

extern "C" int qe_global_avgpool(const float *x, int64_t n_planes, int32_t P, float *out, qe_stream_t stream)
{
    using namespace qe;
    if (n_planes < 0 || P <= 0) return QE_ERR_ARG;
    if (n_planes == 0) return QE_OK;
    if (x == nullptr || out == nullptr) return QE_ERR_ARG;
    if ((size_t)AP_PLANES * P * sizeof(float) > 60 * 1024) return QE_ERR_UNSUPPORTED;   // planes of at most 240 pixels
    const int64_t blocks = (n_planes + AP_PLANES - 1) / AP_PLANES;
    if (blocks > 0x7fffffffLL) return QE_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(global_avgpool_kernel, dim3((unsigned)blocks), dim3(256), (size_t)AP_PLANES * P * sizeof(float),
                       static_cast<hipStream_t>(stream), x, out, n_planes, (int)P);
    QE_LAUNCH_CHECK();
    return QE_OK;
}
