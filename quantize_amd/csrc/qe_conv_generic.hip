// qe_conv_generic.hip -- order-preserving fp32 direct convolution for gfx950.
//
// The general-purpose member of the conv family.  It serves every configuration
// the int8 MFMA kernel (qe_conv_mfma.hip) cannot take exactly: per-input-channel
// activation scale/zero (quantconv2d.cu:115), the float-input operator
// (quantconv2d_float_input.cu), and anything whose integer reformulation is not
// valid.  It keeps the reference's arithmetic literally: operands are
// dequantised to fp32 as ((float)q - zero) * scale and accumulated with one
// fmaf per tap in the reference's ic -> kh -> kw order starting from bias[oc]
// (quantconv2d.cu:92-137), so its result equals the oracle's "fp32_fma" chain.
// Padded taps are zero in the LDS halo instead of being skipped; fmaf(0, w, acc)
// leaves acc unchanged for finite w.
//
// Machine mapping (nothing like the reference's one-thread-per-output kernel):
//   workgroup (256 threads) = one image, PT consecutive output pixels (64 or 256)
//   x OCB = 16*(1024/PT) output channels; each thread owns 4 consecutive pixels and 16 output
//   channels in registers (one LDS weight vector feeds 4 x 16 FMAs: the one-pixel version spent
//   5 LDS reads per 16 FMAs and ran at 5-13 TFLOP/s).  Per chunk of CI input channels the dequantised input
//   band (all rows the tile touches, full padded width) and the dequantised
//   weights [ci][tap][oc] are staged in LDS once and reused by all taps / all
//   threads: input reads are one conflict-free ds_read_b32 per tap, weights are
//   wave-uniform ds_read_b128 broadcasts.
#include "qe_common.h"

namespace qe {

struct GenericConvArgs {
    const void *x;  // packed bytes or fp32
    const float *x_scale, *x_zero;
    int x_bits, x_sign, x_per_tensor;
    const uint8_t *w;
    const float *w_scale, *w_zero;
    int w_bits, w_sign, w_per_tensor;
    const float *bias;
    float *out;
    int N, IC, H, W, OC, KH, KW, stride, padding, OH, OW;
    int PT;           // pixels per tile: 64 or 256
    int CI;           // input channels per LDS chunk
    int IHT_max;      // rows of the LDS input band
    int IWT;          // columns of the LDS input band: W + 2*padding, or (PT-1)*stride + KW in row mode
    int row_mode;     // 1: a tile is PT consecutive pixels of ONE output row and stages only the columns it reads (bands of
                      //    whole padded rows that would not fit the LDS: very wide images)
    int tiles_w;      // row mode: tiles per output row
    int tiles_per_image;
    int oc_blocks;
};

// quantconv2d.cu:103-112 / :118-127 with 64-bit indices.
__device__ __forceinline__ int unpack_elem(const uint8_t *__restrict__ p, int64_t ele_idx, int n_bits, int sign)
{
    const int64_t bit = ele_idx * n_bits;
    const int64_t byte_idx = bit >> 3;
    const int bit_idx = (int)(bit & 7);
    unsigned v = ((unsigned)p[byte_idx] >> bit_idx) & ((1u << n_bits) - 1u);
    if (bit_idx + n_bits > 8) v |= ((unsigned)p[byte_idx + 1] << (8 - bit_idx)) & ((1u << n_bits) - 1u);
    const unsigned offset = sign ? (1u << (n_bits - 1)) : 0u;
    v = (v - offset) & 0xffu;
    return sign ? (int)(int8_t)v : (int)v;
}

constexpr int GC_THREADS = 256;
constexpr int GC_OCT = 16;  // output channels per thread
constexpr int GC_PXT = 4;   // consecutive output pixels per thread: a weight vector read from LDS feeds 4 x 16 FMAs

template <bool PACKED_IN>
__global__ __launch_bounds__(GC_THREADS) void conv_generic_kernel(const GenericConvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int KK = a.KH * a.KW;
    const int TPX = a.PT / GC_PXT;                   // threads along the pixels
    const int OCB = GC_OCT * (GC_THREADS / TPX);
    float *Xs = smem;                                          // [CI][IHT_max][IWT]
    float *Ws = smem + (size_t)a.CI * a.IHT_max * a.IWT;       // [CI][KK][OCB]
    // keep Ws 16-byte aligned for the float4 reads
    Ws = reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(Ws) + 15) & ~(uintptr_t)15);

    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    const int ocb = bid % a.oc_blocks; bid /= a.oc_blocks;
    const int tile = bid % a.tiles_per_image;
    const int n = bid / a.tiles_per_image;

    const int OHW = a.OH * a.OW;
    int p0, p_end, oh_first, oh_last, iw_base;               // pixels [p0, p_end) of the image, first staged input column
    if (a.row_mode) {
        oh_first = oh_last = tile / a.tiles_w;
        const int seg = tile - oh_first * a.tiles_w;
        p0 = oh_first * a.OW + seg * a.PT;
        p_end = oh_first * a.OW + min(a.OW, (seg + 1) * a.PT);
        iw_base = seg * a.PT * a.stride - a.padding;
    } else {
        p0 = tile * a.PT;
        p_end = min(p0 + a.PT, OHW);
        oh_first = p0 / a.OW;
        oh_last = (p_end - 1) / a.OW;
        iw_base = -a.padding;
    }
    const int ow_base = a.row_mode ? p0 - oh_first * a.OW : 0;   // first output column the LDS band is aligned to
    const int ih0 = oh_first * a.stride - a.padding;
    const int IHT = (oh_last - oh_first) * a.stride + a.KH;  // <= IHT_max

    const int pxt = tid % TPX;
    const int ocg = tid / TPX;
    int pj[GC_PXT], xoff[GC_PXT];
    bool pv[GC_PXT];
#pragma unroll
    for (int j = 0; j < GC_PXT; ++j) {
        const int p = p0 + GC_PXT * pxt + j;
        pv[j] = p < p_end;
        const int oh = pv[j] ? p / a.OW : oh_first;
        const int ow = pv[j] ? p % a.OW : ow_base;
        pj[j] = p;
        xoff[j] = ((oh - oh_first) * a.stride) * a.IWT + (ow - ow_base) * a.stride;   // + kh * IWT + kw (LDS column = iw - iw_base)
    }
    const int oc0 = ocb * OCB + ocg * GC_OCT;

    float acc[GC_PXT][GC_OCT];
#pragma unroll
    for (int i = 0; i < GC_OCT; ++i) {
        const int oc = oc0 + i;
        const float b = (a.bias != nullptr && oc < a.OC) ? a.bias[oc] : 0.0f;  // quantconv2d.cu:92
#pragma unroll
        for (int j = 0; j < GC_PXT; ++j) acc[j][i] = b;
    }

    const int64_t x_img = (int64_t)n * a.IC * a.H * a.W;

    for (int c0 = 0; c0 < a.IC; c0 += a.CI) {
        const int cn = min(a.CI, a.IC - c0);
        // ---- stage the dequantised input band -------------------------------------------
        const int band = IHT * a.IWT;
        for (int idx = tid; idx < cn * band; idx += GC_THREADS) {
            const int ci = idx / band;
            const int rem = idx - ci * band;
            const int ihl = rem / a.IWT;
            const int iwl = rem - ihl * a.IWT;
            const int ih = ih0 + ihl, iw = iwl + iw_base;
            float v = 0.0f;
            if (ih >= 0 && ih < a.H && iw >= 0 && iw < a.W) {  // quantconv2d.cu:101
                const int ic = c0 + ci;
                const int64_t e = x_img + ((int64_t)ic * a.H + ih) * a.W + iw;
                if constexpr (PACKED_IN) {
                    const int q = unpack_elem(static_cast<const uint8_t *>(a.x), e, a.x_bits, a.x_sign);
                    const float zx = a.x_per_tensor ? a.x_zero[0] : a.x_zero[ic];
                    const float sx = a.x_per_tensor ? a.x_scale[0] : a.x_scale[ic];
                    v = ((float)q - zx) * sx;                  // quantconv2d.cu:113-115
                } else {
                    v = static_cast<const float *>(a.x)[e];    // quantconv2d_float_input.cu:109
                }
            }
            Xs[(ci * a.IHT_max + ihl) * a.IWT + iwl] = v;
        }
        // ---- stage the dequantised weights [ci][tap][oc] --------------------------------
        for (int idx = tid; idx < cn * KK * OCB; idx += GC_THREADS) {
            const int o = idx % OCB;
            const int rem = idx / OCB;
            const int tap = rem % KK;
            const int ci = rem / KK;
            const int oc = ocb * OCB + o;
            float v = 0.0f;
            if (oc < a.OC) {
                const int ic = c0 + ci;
                const int64_t e = ((int64_t)oc * a.IC + ic) * KK + tap;  // quantconv2d.cu:118
                const int q = unpack_elem(a.w, e, a.w_bits, a.w_sign);
                const float zw = a.w_per_tensor ? a.w_zero[0] : a.w_zero[oc];
                const float sw = a.w_per_tensor ? a.w_scale[0] : a.w_scale[oc];
                v = ((float)q - zw) * sw;                                // quantconv2d.cu:128-130
            }
            Ws[(ci * KK + tap) * OCB + o] = v;
        }
        __syncthreads();

        // ---- accumulate: ic -> kh -> kw, one fmaf per tap (quantconv2d.cu:95-137) --------
        for (int ci = 0; ci < cn; ++ci) {
            const float *xch = Xs + (ci * a.IHT_max) * a.IWT;
            const float *wrow = Ws + (ci * KK) * OCB + ocg * GC_OCT;
            for (int kh = 0; kh < a.KH; ++kh) {
                for (int kw = 0; kw < a.KW; ++kw) {
                    float xv[GC_PXT];
#pragma unroll
                    for (int j = 0; j < GC_PXT; ++j) xv[j] = xch[xoff[j] + kh * a.IWT + kw];
                    const float4 *w4 = reinterpret_cast<const float4 *>(wrow + (kh * a.KW + kw) * OCB);
#pragma unroll
                    for (int q = 0; q < GC_OCT / 4; ++q) {
                        const float4 wv = w4[q];
#pragma unroll
                        for (int j = 0; j < GC_PXT; ++j) {
                            acc[j][4 * q + 0] = fmaf(xv[j], wv.x, acc[j][4 * q + 0]);
                            acc[j][4 * q + 1] = fmaf(xv[j], wv.y, acc[j][4 * q + 1]);
                            acc[j][4 * q + 2] = fmaf(xv[j], wv.z, acc[j][4 * q + 2]);
                            acc[j][4 * q + 3] = fmaf(xv[j], wv.w, acc[j][4 * q + 3]);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }

#pragma unroll
    for (int j = 0; j < GC_PXT; ++j) {
        if (!pv[j]) continue;
        float *o = a.out + ((int64_t)n * a.OC) * OHW + pj[j];
#pragma unroll
        for (int i = 0; i < GC_OCT; ++i) {
            const int oc = oc0 + i;
            if (oc < a.OC) o[(int64_t)oc * OHW] = acc[j][i];  // quantconv2d.cu:140
        }
    }
}

// Host-side tiling choice + launch.  Returns QE_* status.
int launch_conv_generic(bool packed_in, const void *x, const qe_qparam *xq, const qe_qparam *w,
                        const float *bias, const qe_conv_shape *sh, float *out, hipStream_t s)
{
    GenericConvArgs a;
    a.x = x;
    a.x_scale = xq ? xq->scale : nullptr;
    a.x_zero = xq ? xq->zero : nullptr;
    a.x_bits = xq ? xq->n_bits : 0;
    a.x_sign = xq ? xq->sign : 0;
    a.x_per_tensor = xq ? (xq->n_param == 1) : 1;
    a.w = w->data; a.w_scale = w->scale; a.w_zero = w->zero;
    a.w_bits = w->n_bits; a.w_sign = w->sign; a.w_per_tensor = (w->n_param == 1);
    a.bias = bias; a.out = out;
    a.N = sh->N; a.IC = sh->IC; a.H = sh->H; a.W = sh->W;
    a.OC = sh->OC; a.KH = sh->KH; a.KW = sh->KW; a.stride = sh->stride; a.padding = sh->padding;
    a.OH = (sh->H + 2 * sh->padding - sh->KH) / sh->stride + 1;
    a.OW = (sh->W + 2 * sh->padding - sh->KW) / sh->stride + 1;
    if (a.OH <= 0 || a.OW <= 0 || a.N == 0 || a.OC == 0) return QE_OK;  // empty output

    const int OHW = a.OH * a.OW;
    a.PT = (OHW <= 64) ? 64 : 256;
    const int OCB = GC_OCT * (GC_THREADS / (a.PT / GC_PXT));
    a.IWT = a.W + 2 * a.padding;
    const int rows_max = min(a.OH, (a.PT - 1) / a.OW + 2);
    a.IHT_max = (rows_max - 1) * a.stride + a.KH;
    const int KK = a.KH * a.KW;
    const size_t per_ic = ((size_t)a.IHT_max * a.IWT + (size_t)KK * OCB) * sizeof(float);
    const size_t budget = 60 * 1024;
    size_t per = per_ic;
    a.row_mode = 0; a.tiles_w = 1;
    if (budget / per < 1) {
        // one input channel's band of whole padded rows does not fit the LDS (a 7x7 conv on ~1500-pixel-wide images):
        // tiles of PT pixels inside ONE output row, staging only the (PT-1)*stride + KW columns they read
        a.row_mode = 1;
        a.tiles_w = (a.OW + a.PT - 1) / a.PT;
        a.IWT = (a.PT - 1) * a.stride + a.KW;
        a.IHT_max = a.KH;
        per = ((size_t)a.IHT_max * a.IWT + (size_t)KK * OCB) * sizeof(float);
        if (budget / per < 1) return QE_ERR_UNSUPPORTED;   // (kernels of hundreds of taps: weights of one channel alone exceed the LDS)
    }
    int ci = (int)(budget / per);
    if (ci > a.IC) ci = a.IC;
    if (ci > 64) ci = 64;
    a.CI = ci;
    a.tiles_per_image = a.row_mode ? a.OH * a.tiles_w : (OHW + a.PT - 1) / a.PT;
    a.oc_blocks = (a.OC + OCB - 1) / OCB;
    const int64_t blocks = (int64_t)a.N * a.tiles_per_image * a.oc_blocks;
    if (blocks > 0x7fffffffLL) return QE_ERR_UNSUPPORTED;
    const size_t shmem = (size_t)a.CI * per + 16;
    if (packed_in)
        hipLaunchKernelGGL(conv_generic_kernel<true>, dim3((unsigned)blocks), dim3(GC_THREADS), shmem, s, a);
    else
        hipLaunchKernelGGL(conv_generic_kernel<false>, dim3((unsigned)blocks), dim3(GC_THREADS), shmem, s, a);
    QE_LAUNCH_CHECK();
    return QE_OK;
}

}  // namespace qe
