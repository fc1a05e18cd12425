/*
 * quant_engine.h -- C ABI of the MI355X (gfx950) quantized-conv2d engine.
 *
 * This is the drop-in boundary for the reference's `engine.kernels` extension
 * (JingInAI/Quantize, pybind module `quant_engine`, engine/kernels/pybind.cpp:7-17).
 * Every entry point takes plain device pointers, sizes and a HIP stream; no torch
 * types appear here.  The torch-facing `quant_engine` Python module
 * (quantize_amd/csrc/torch_binding.cpp) is a thin layer over these calls that
 * reproduces the reference's argument checks, allocation and error class.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless stated otherwise;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *     every call is asynchronous on that stream and performs no host sync,
 *     allocation or free (graph-capturable);
 *   - return value: QE_OK or a QE_ERR_* code; qe_error_string() gives the text,
 *     which for the reference's own checks is the reference's message verbatim;
 *   - packed tensor format (reference: engine/kernels/tpack/tpack.cu:50-81,
 *     tpack.h:14-15): element i of an n_bits-wide tensor occupies bits
 *     [i*n_bits, (i+1)*n_bits) of a little-endian bit stream, stored value =
 *     q + (sign ? 2^(n_bits-1) : 0); the stream is ceil(n*n_bits/8) bytes, no
 *     padding between elements, rows or channels.
 */
#ifndef QUANT_ENGINE_H
#define QUANT_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *qe_stream_t; /* hipStream_t */

enum qe_status {
    QE_OK = 0,
    QE_ERR_NBITS = 1,       /* "n_bits must be in the range (0, 8]"  tpack.cu:13 */
    QE_ERR_RANGE = 2,       /* "The input tensor is out of range."   tpack.cu:14 (reported through `status`, see qe_tpack) */
    QE_ERR_DTYPE = 3,       /* dtype code not in enum qe_dtype */
    QE_ERR_ARG = 4,         /* null pointer / negative size / inconsistent shape */
    QE_ERR_HIP = 5,         /* a HIP runtime call failed; see qe_last_hip_error() */
    QE_ERR_WORKSPACE = 6,   /* workspace too small */
    QE_ERR_UNSUPPORTED = 7  /* shape outside what the kernels index */
};

/* Input element types accepted by qe_tpack: the set the reference dispatches on
 * (AT_DISPATCH_ALL_TYPES_AND(Half), tpack.cu:120). */
enum qe_dtype {
    QE_U8 = 0, QE_I8 = 1, QE_I16 = 2, QE_I32 = 3, QE_I64 = 4,
    QE_F16 = 5, QE_F32 = 6, QE_F64 = 7
};

const char *qe_error_string(int status);
/* hipError_t of the last failing HIP call made by this library on this thread (0 = none). */
int qe_last_hip_error(void);
/* Library version, and the gfx target the device code was built for ("gfx950"). */
const char *qe_version(void);
const char *qe_target_arch(void);

/* ceil(n_elements * n_bits / 8): size of the packed byte stream (tpack.cu:224). */
int64_t qe_packed_nbytes(int64_t n_elements, int n_bits);

/* ---------------------------------------------------------------------------
 * qe_tpack -- replaces tpack()/tpack_cuda()/tpack_cuda_kernel
 *   reference: engine/kernels/tpack/tpack.cu:203-255, :96-128, :30-84.
 * x        n elements of `dtype`, contiguous.
 * out      qe_packed_nbytes(n, n_bits) bytes; EVERY byte is written (no
 *          pre-zeroing needed, unlike tpack.cu:225).
 * status   optional device int32[1], must be zero on entry; bit 0 is set when any
 *          element fails the reference's range check (tpack.cu:211-215: value,
 *          as float, outside [-2^(b-1), 2^(b-1)-1] (sign) or [0, 2^b-1], or NaN).
 *          The caller reads it back to raise "The input tensor is out of
 *          range."; `out` is unspecified in that case (the reference raises
 *          before packing).
 * ------------------------------------------------------------------------- */
int qe_tpack(const void *x, int dtype, int64_t n, int n_bits, int sign,
             uint8_t *out, int32_t *status, qe_stream_t stream);

/* ---------------------------------------------------------------------------
 * qe_tunpack -- replaces tunpack()/tunpack_cuda()/tunpack_cuda_kernel
 *   reference: engine/kernels/tpack/tpack.cu:429-476, :327-359, :267-315.
 * packed   qe_packed_nbytes(n, n_bits) bytes.
 * out      n bytes: int8 when sign, uint8 otherwise (tpack.cu:452-455).
 * ------------------------------------------------------------------------- */
int qe_tunpack(const uint8_t *packed, int64_t n, int n_bits, int sign,
               void *out, qe_stream_t stream);

/* ---------------------------------------------------------------------------
 * qe_quantize_pack -- Quantizer + tpack in one pass
 *   reference: modelzoo/modules/quantizer.py:31,213-226 (q = round(x / scale - zero).clamp(qmin, qmax), returned as an
 *   integer-valued fp32 tensor in packed mode) followed by engine.tpack (tpack.cu:203-255).
 * The packed conv / linear operators have no producer for their activation operand in the reference (the module
 * hands fp32 q to F.conv2d); this is that producer: fp32 activations in, the b-bit stream qe_quantconv2d takes out,
 * without the 4 B/element intermediate.  out is bit-identical to qe_tpack(round(x / scale - zero).clamp(..)).
 * scale/zero  n_param fp32 elements in the MODULE's convention (the value is x / scale - zero; the conv kernels'
 *             (q - zero') dequantisation takes zero' = -zero).  n_param == 1: per tensor; otherwise per channel with
 *             channel(i) = (i / inner) % n_param  (NCHW activations: inner = H*W, n_param = C).
 * status      as qe_tpack: bit 0 set when a clamped value does not fit n_bits/sign (or is NaN).
 * ------------------------------------------------------------------------- */
int qe_quantize_pack(const float *x, int64_t n, const float *scale, const float *zero, int32_t n_param,
                     int64_t inner, float qmin, float qmax, int n_bits, int sign, uint8_t *out,
                     int32_t *status, qe_stream_t stream);

/* ---------------------------------------------------------------------------
 * Convolution problem description shared by the two conv entry points.
 * Shapes follow the reference's host code: square stride/padding, no dilation,
 * no groups (functions/quantconv2d.cu:198-211).  OH/OW are derived:
 * OH = (H + 2*padding - KH)/stride + 1.
 * ------------------------------------------------------------------------- */
typedef struct qe_conv_shape {
    int32_t N, IC, H, W;     /* input  (N, IC, H, W)   NCHW */
    int32_t OC, KH, KW;      /* weight (OC, IC, KH, KW) OIHW */
    int32_t stride, padding;
} qe_conv_shape;

/* Quantisation parameters of one packed operand.
 * scale/zero: device fp32 arrays of `n_param` elements.  n_param == 1 means per
 * tensor (the reference's `numel() == 1` test, quantconv2d.cu:238,244);
 * otherwise activations index by INPUT channel (quantconv2d.cu:115) and weights
 * by OUTPUT channel (quantconv2d.cu:130).  Dequantisation follows the kernel
 * convention (q - zero) * scale (quantconv2d.cu:113-115,128-130). */
typedef struct qe_qparam {
    const uint8_t *data;   /* packed bit stream */
    int32_t n_bits;        /* 1..8  (des[0]) */
    int32_t sign;          /* 0/1   (des[1]) */
    const float *scale;
    const float *zero;
    int32_t n_param;
} qe_qparam;

/* Bytes of scratch qe_quantconv2d needs for this problem (0 is possible).
 * The scratch holds the re-laid-out int8 weights and per-channel epilogue
 * constants of the MFMA path; contents are dead after the call returns to the
 * stream order (i.e. may be reused by the next call on the same stream). */
size_t qe_quantconv2d_workspace_bytes(const qe_conv_shape *shape, int x_bits, int w_bits);

/* ---------------------------------------------------------------------------
 * qe_quantconv2d -- replaces quantconv2d()/quantconv2d_cuda_kernel
 *   reference: engine/kernels/functions/quantconv2d.cu:164-264, :49-142.
 * out[n,oc,oh,ow] = bias[oc] + sum over in-bounds taps (ic,kh,kw) of
 *     ((qx - zx) * sx) * ((qw - zw) * sw)          (fp32 result, NCHW, contiguous)
 * x, w     packed operands (see qe_qparam); x holds N*IC*H*W elements, w holds
 *          OC*IC*KH*KW elements.
 * bias     fp32[OC] or NULL.
 * out      fp32[N*OC*OH*OW]; every element is written.
 * workspace/workspace_bytes  scratch of at least qe_quantconv2d_workspace_bytes().
 * ------------------------------------------------------------------------- */
int qe_quantconv2d(const qe_qparam *x, const qe_qparam *w, const float *bias,
                   const qe_conv_shape *shape, float *out,
                   void *workspace, size_t workspace_bytes, qe_stream_t stream);

/* ---------------------------------------------------------------------------
 * Weights kept prepared across calls.
 *   reference: modelzoo/modules/quantconv2d.py:187-192 (pack() stores the packed weight once) and :230-233 (the
 *   TODO "remove tunpack in loading state_dict, and add support for custom operators"): a packed layer's weights,
 *   scales and bias never change between forward passes, yet the reference op receives them packed on every call.
 * qe_conv_prepare runs the x-independent part of qe_quantconv2d once -- the packed OIHW stream re-laid out as
 * int8 MFMA fragments, per-channel (sw, zw', bias) and the border-aware tap-sum tables -- into a caller-owned
 * device buffer of qe_conv_prepared_bytes() bytes; qe_quantconv2d_prepared then runs only the convolution.
 * Results are bit-identical to qe_quantconv2d.  x_bits selects the kernel plan the tables are laid out for and
 * must match the activations passed later; w / bias must be the same tensors (the 1x1 kernels still read them).
 * qe_conv_prepared_bytes() == 0: nothing to prepare (the kernel consumes the packed weights directly); both
 * calls then accept prepared == NULL.  Scratch for the prepared call: qe_quantconv2d_prepared_workspace_bytes().
 * ------------------------------------------------------------------------- */
size_t qe_conv_prepared_bytes(const qe_conv_shape *shape, int x_bits, int w_bits);
/* Signature of the prepared tables' layout (0: nothing to prepare).  It depends on the weight tensor's geometry and on
 * the kernel family the plan picks, NOT on the batch size, and on the image size only where that changes the family:
 * a caller that keeps prepared buffers per layer keys them on this value, so alternating batch sizes (or image sizes
 * served by the same family) reuse one buffer instead of re-preparing. */
uint64_t qe_conv_prepared_layout(const qe_conv_shape *shape, int x_bits, int w_bits);
size_t qe_quantconv2d_prepared_workspace_bytes(const qe_conv_shape *shape, int x_bits, int w_bits);
int qe_conv_prepare(const qe_qparam *w, const float *bias, const qe_conv_shape *shape, int x_bits,
                    void *prepared, size_t prepared_bytes, qe_stream_t stream);
int qe_quantconv2d_prepared(const qe_qparam *x, const qe_qparam *w, const float *bias,
                            const qe_conv_shape *shape, const void *prepared, size_t prepared_bytes,
                            float *out, void *workspace, size_t workspace_bytes, qe_stream_t stream);

/* ---------------------------------------------------------------------------
 * qe_quantconv2d_requant_prepared -- quantconv2d with the NEXT layer's activation quantiser fused into the epilogue
 *   (conv-epilogue form).  reference: the packed forward of
 *   modelzoo/modules/quantconv2d.py:198-210 followed by the consumer's Quantizer (quantizer.py:31,213-226) and
 *   engine.tpack -- three passes over a 4 B/element tensor there, none here.
 * out  qe_packed_nbytes(N*OC*OH*OW, rq->n_bits) bytes, bit-identical to
 *      qe_quantize_pack(qe_quantconv2d_prepared(x, w, ...), rq->scale, rq->zero, rq->n_param, OH*OW, ...).
 * rq   output quantiser in the MODULE's convention (q = round(y / scale - zero).clamp(qmin, qmax)); n_param = 1 (per
 *      tensor) or OC (per output channel).
 * qe_quantconv2d_requant_path: 1 = the conv kernel stores the codes itself (8-bit codes, per-tensor rq, MFMA-eligible problem);
 *      0 = two passes inside the call (fp32 y in the workspace, then the quantise+pack kernel).
 * workspace  qe_quantconv2d_requant_workspace_bytes(shape, x, w, rq) bytes, 16-byte aligned.
 * prepared   as qe_quantconv2d_prepared (qe_conv_prepare); status as qe_tpack.
 * ------------------------------------------------------------------------- */
typedef struct qe_requant {
    const float *scale;
    const float *zero;
    int32_t n_param;
    float qmin, qmax;
    int32_t n_bits, sign;
} qe_requant;
int qe_quantconv2d_requant_path(const qe_conv_shape *shape, const qe_qparam *x, const qe_qparam *w, const qe_requant *rq);
size_t qe_quantconv2d_requant_workspace_bytes(const qe_conv_shape *shape, const qe_qparam *x, const qe_qparam *w,
                                              const qe_requant *rq);
int qe_quantconv2d_requant_prepared(const qe_qparam *x, const qe_qparam *w, const float *bias,
                                    const qe_conv_shape *shape, const void *prepared, size_t prepared_bytes,
                                    const qe_requant *rq, uint8_t *out, int32_t *status,
                                    void *workspace, size_t workspace_bytes, qe_stream_t stream);

/* ---------------------------------------------------------------------------
 * qe_quantconv2d_float_input -- replaces quantconv2d_float_input()/..._cuda
 *   reference: engine/kernels/functions/quantconv2d_float_input.cu:140-220, :45-121.
 * x        fp32[N*IC*H*W], NCHW contiguous.
 * ------------------------------------------------------------------------- */
int qe_quantconv2d_float_input(const float *x, const qe_qparam *w, const float *bias,
                               const qe_conv_shape *shape, float *out, qe_stream_t stream);

/* The same operator on the matrix cores (exact 3-way bf16 split of the fp32 activations x integer weight codes,
 * fp32 accumulation; quantize_amd/csrc/qe_conv_f32.hip).  It needs scratch for the weights in fragment order, which
 * the reference's signature has no place for, hence the _ws form; the plain entry point above keeps the
 * order-preserving VALU kernel (bit-identical to the reference's fmaf chain).  Results of the two differ by fp32
 * accumulation order only (within max(1e-5, |reference chain - exact|), tests/test_conv_f32_gpu.py).
 * qe_quantconv2d_float_input_workspace_bytes() == 0 or path == 0: the problem stays on the VALU kernel.
 * The weight tables are x-independent: qe_conv_f32_prepare once + qe_quantconv2d_float_input_prepared per call keeps
 * them across forward passes (same contract as qe_conv_prepare). */
size_t qe_quantconv2d_float_input_workspace_bytes(const qe_conv_shape *shape, int w_bits);
int qe_quantconv2d_float_input_ws(const float *x, const qe_qparam *w, const float *bias,
                                  const qe_conv_shape *shape, float *out, void *workspace,
                                  size_t workspace_bytes, qe_stream_t stream);
int qe_conv_f32_prepare(const qe_qparam *w, const float *bias, const qe_conv_shape *shape,
                        void *prepared, size_t prepared_bytes, qe_stream_t stream);
int qe_quantconv2d_float_input_prepared(const float *x, const qe_qparam *w, const float *bias,
                                        const qe_conv_shape *shape, const void *prepared,
                                        size_t prepared_bytes, float *out, qe_stream_t stream);
/* 0 = order-preserving VALU kernel, 1 = bf16 MFMA kernel */
int qe_quantconv2d_float_input_path(const qe_conv_shape *shape, const qe_qparam *w);

/* Which kernel family qe_quantconv2d will pick for a problem (for tests, bench
 * and profiles): 0 = generic fp32 direct convolution, 1 = int8 MFMA implicit GEMM. */
int qe_quantconv2d_path(const qe_conv_shape *shape, const qe_qparam *x, const qe_qparam *w);

/* ---- quantlinear ------------------------------------------------------------
 * Replaces quantlinear / quantlinear_cuda (functions/quantlinear.cu:233-297, :153-214):
 *   out[b, o] = bias[o] + sum_k (q_x[b,k] + zx[b]) * (q_w[o,k] + zw[o]) * (sx[b] * sw[o])
 * Both operands are packed b-bit streams (x: B*K elements, row major; w: O*K elements, OIHW of a Linear).
 * NOTE the conventions of THIS reference kernel (they differ from quantconv2d): the zero point is ADDED
 * (quantlinear.cu:115,120) and the activation scale/zero are indexed by the batch ROW (:96-99).
 * x->n_param is 1 (broadcast: the reference expands 0-dim tensors, :276-282) or B; w->n_param is 1 or O.
 * bias may be NULL (the reference substitutes zeros, :268).  out: B*O floats, fully written.
 * The reference accumulates stale shared memory when K % 32 != 0 (:76-92 never zero-fills the tail);
 * this entry point computes the sum over k < K for every K.                                            */
int qe_quantlinear(const qe_qparam *x, const qe_qparam *w, const float *bias,
                   int64_t B, int32_t K, int32_t O, float *out, qe_stream_t stream);

/* Replaces quantlinear_float_input (functions/quantlinear_float_input.cu:120-182, kernel :36-104):
 *   out[b, o] = bias[o] + sum_k x[b,k] * ((q_w[o,k] - zw[o]) * sw[o])      ((q - zero): :82-86)
 * x: B*K floats.  w->n_param is 1 (per tensor iff numel()==1, :170) or O.
 * Non-finite activations (this entry point and qe_quantconv2d_float_input, on their MFMA kernels): every output that depends
 * on a +-inf or NaN activation is non-finite -- NaN where the reference's fmaf chain may keep +-inf (the exact bf16 split of
 * an infinity leaves inf - inf in the remainder terms) -- and every other output is unaffected.                          */
int qe_quantlinear_float_input(const float *x, const qe_qparam *w, const float *bias,
                               int64_t B, int32_t K, int32_t O, float *out, qe_stream_t stream);

/* 0 = order-preserving fp32 kernel, 1 = int8 MFMA GEMM (8-bit x 8-bit operands, K % 64 == 0, 16-byte aligned streams). */
int qe_quantlinear_path(const qe_qparam *x, const qe_qparam *w, int64_t B, int32_t K, int32_t O);

/* quantlinear_float_input: 0 = order-preserving fp32 kernel (bit-identical to the reference's fused chain), 1 = bf16 MFMA
 * GEMM on an exact three-way split of the activations (8-bit weights, K % 32 == 0, 16-byte aligned operands; results within
 * fp32 accumulation rounding of the exact sum; QE_LIN_F32_MFMA=0 disables it). */
int qe_quantlinear_float_input_path(const float *x, const qe_qparam *w, int64_t B, int32_t K, int32_t O);

/* ---- auxiliary (no counterpart in the reference's extension) ---------------------------------
 * Global average pool of an fp32 NCHW tensor: out[plane] = mean(x[plane][0..P)) for n_planes = N*C planes of P
 * contiguous floats.  The reference's models do this in PyTorch (torchvision ResNet: AdaptiveAvgPool2d); bench.py's
 * top-1 tail uses this entry point because torch's reduction reads the (256,2048,7,7) conv output at 1.5 TB/s.     */
int qe_global_avgpool(const float *x, int64_t n_planes, int32_t P, float *out, qe_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* QUANT_ENGINE_H */
