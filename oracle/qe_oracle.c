/*
 * qe_oracle.c -- CPU restatement of the reference `engine.kernels` hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (quantize_amd/, the
 * C-ABI library, the `quant_engine` module) may import, link or call this file.
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg use it,
 * and only as the checker / the timed CPU baseline.
 *
 * Every function cites the reference lines it restates (paths relative to
 * /root/reference).  The loops follow the reference statement by statement:
 * same element order, same integer types (unsigned char wrap-around), fp32
 * arithmetic with one rounding per operation.  Deliberate differences, all
 * documented in DESIGN.md:
 *   - 64-bit element/bit indices (the reference uses 32-bit `int`/`unsigned`
 *     and silently overflows beyond 2^31 bits / 2^32 element-bits);
 *   - plain C arrays instead of torch tensors.
 *
 * Parity pinning: see oracle/README.md.  tpack/tunpack are pinned bit-exactly
 * by golden vectors generated from the reference's own Python packer
 * (engine/utils/tensor_packing.py); the conv functions are pinned against the
 * reference's packed-forward fallback (F.conv2d on dequantised tensors,
 * modelzoo/modules/quantconv2d.py:207-210) and a captured QuantConv2d
 * calibrate -> pack -> reload -> forward run of the reference modules.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 * -ffp-contract=off matters: the mul+add chain must not be fused behind our
 * back; the fused variant calls fmaf() explicitly.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#define QE_ORACLE_OK 0
#define QE_ORACLE_ERR_NBITS 1 /* "n_bits must be in the range (0, 8]"   tpack.cu:13  */
#define QE_ORACLE_ERR_RANGE 2 /* "The input tensor is out of range."    tpack.cu:14  */

/* ------------------------------------------------------------------------- */
/* tpack: engine/kernels/tpack/tpack.cu:140-190 (tpack_cpu) with the host     */
/* checks of tpack.cu:203-225 (CHECK_NBITS, CHECK_RANGE, ceil(n*b/8) zeros).  */
/* `x` is the tensor already converted to float, exactly as tpack_cpu reads   */
/* it through x[i].item<float>() (tpack.cu:157).                              */
/* out must hold (n*n_bits+7)/8 bytes; it is zeroed here (tpack.cu:225).      */
/* ------------------------------------------------------------------------- */
int qe_oracle_tpack(const float *x, int64_t n, int n_bits, int sign, uint8_t *out)
{
    if (!(n_bits > 0 && n_bits <= 8)) /* tpack.cu:13,209 */
        return QE_ORACLE_ERR_NBITS;

    /* tpack.cu:211-215: x.min()/x.max() as float against the integer bounds.
     * A NaN makes both comparisons false -> error, like TORCH_CHECK. */
    if (n > 0) {
        float xmin = x[0], xmax = x[0];
        int has_nan = 0;
        for (int64_t i = 0; i < n; i++) {
            if (x[i] != x[i]) has_nan = 1;
            if (x[i] < xmin) xmin = x[i];
            if (x[i] > xmax) xmax = x[i];
        }
        int lo = sign ? -(1 << (n_bits - 1)) : 0;
        int hi = sign ? (1 << (n_bits - 1)) - 1 : (1 << n_bits) - 1;
        if (has_nan || !(xmin >= (float)lo && xmax <= (float)hi))
            return QE_ORACLE_ERR_RANGE;
    }

    if (n < 0) n = 0;
    uint64_t n_out = ((uint64_t)n * (uint64_t)n_bits + 7u) / 8u; /* tpack.cu:224 */
    memset(out, 0, (size_t)n_out);

    unsigned char offset = 0; /* tpack.cu:148-151 */
    if (sign) offset = (unsigned char)(1 << (n_bits - 1));

    for (int64_t i = 0; i < n; i++) { /* tpack.cu:154-189 */
        unsigned char element = (unsigned char)(signed char)x[i]; /* :157 (char)float */
        element += offset;                                        /* :158 */
        int64_t bit_index = i * n_bits;                           /* :161 */
        int64_t byte_index = bit_index / 8;                       /* :164 */
        int bit_offset = (int)(bit_index % 8);                    /* :167 */
        unsigned char byte = out[byte_index];                     /* :170 */
        byte |= (unsigned char)(element << bit_offset);           /* :173 */
        out[byte_index] = byte;                                   /* :176 */
        if (bit_offset + n_bits > 8) {                            /* :178 */
            unsigned char b2 = out[byte_index + 1];               /* :181 */
            b2 |= (unsigned char)(element >> (8 - bit_offset));   /* :184 */
            out[byte_index + 1] = b2;                             /* :187 */
        }
    }
    return QE_ORACLE_OK;
}

/* ------------------------------------------------------------------------- */
/* tunpack: engine/kernels/tpack/tpack.cu:371-419 (tunpack_cpu).              */
/* out holds n bytes: the int8 (sign) or uint8 (!sign) values, bit-for-bit    */
/* (tpack.cu:413-417 stores (signed char)element or (unsigned char)element:   */
/* the same byte).                                                            */
/* ------------------------------------------------------------------------- */
int qe_oracle_tunpack(const uint8_t *in, int64_t n, int n_bits, int sign, uint8_t *out)
{
    if (!(n_bits > 0 && n_bits <= 8)) /* tpack.cu:438 */
        return QE_ORACLE_ERR_NBITS;
    unsigned char offset = 0; /* tpack.cu:379-382 */
    if (sign) offset = (unsigned char)(1 << (n_bits - 1));
    for (int64_t i = 0; i < n; i++) { /* tpack.cu:385-418 */
        int64_t bit_index = i * n_bits;
        int64_t byte_index = bit_index / 8;
        int bit_offset = (int)(bit_index % 8);
        unsigned char byte = in[byte_index];
        unsigned char element = (unsigned char)((byte >> bit_offset) & ((1 << n_bits) - 1)); /* :400 */
        if (bit_offset + n_bits > 8) {                                                       /* :402 */
            unsigned char b2 = in[byte_index + 1];
            element |= (unsigned char)((b2 << (8 - bit_offset)) & ((1 << n_bits) - 1));      /* :408 */
        }
        element -= offset; /* :412 */
        out[i] = element;
    }
    return QE_ORACLE_OK;
}

/* One packed element, as the conv kernels read it:
 * engine/kernels/functions/quantconv2d.cu:103-112 and :118-127.
 * Returns the value after `value -= offset; (T)value` as an int
 * (T = int8_t when sign, uint8_t otherwise: quantconv2d.cu:226-227). */
static inline int qe_unpack_elem(const uint8_t *p, int64_t ele_idx, int n_bits, int sign)
{
    int64_t byte_idx = ele_idx * n_bits / 8;
    int bit_idx = (int)(ele_idx * n_bits % 8);
    unsigned char v = (unsigned char)((p[byte_idx] >> bit_idx) & ((1 << n_bits) - 1));
    if (bit_idx + n_bits > 8)
        v |= (unsigned char)((p[byte_idx + 1] << (8 - bit_idx)) & ((1 << n_bits) - 1));
    unsigned char offset = sign ? (unsigned char)(1 << (n_bits - 1)) : 0; /* :218-219 */
    v -= offset;
    return sign ? (int)(signed char)v : (int)v;
}

/* ------------------------------------------------------------------------- */
/* quantconv2d: engine/kernels/functions/quantconv2d.cu:78-141 (one output    */
/* element per "thread", loops ic -> kh -> kw, padded taps skipped :101) with  */
/* the host-side shape arithmetic of :210-211.                                */
/*   mode 0: fp32, product then add (the source as written, :133)             */
/*   mode 1: fp32, fmaf(x, w, acc) (what nvcc's default -fmad=true emits)     */
/*   mode 2: double accumulation of the double-precision dequantised product  */
/*           ("exact" reference for tolerance bookkeeping), rounded to fp32   */
/*           only at the end (out_f64 receives the unrounded double if given) */
/* x_scale/x_zero: 1 element (per tensor, :238) or indexed by input channel   */
/* (:115); w_scale/w_zero: 1 element or indexed by output channel (:130).     */
/* ------------------------------------------------------------------------- */
int qe_oracle_quantconv2d(
    const uint8_t *x, int x_bits, int x_sign,
    const float *x_scale, const float *x_zero, int x_per_tensor,
    const uint8_t *w, int w_bits, int w_sign,
    const float *w_scale, const float *w_zero, int w_per_tensor,
    const float *bias, /* may be NULL */
    int N, int IC, int H, int W, int OC, int KH, int KW, int stride, int padding,
    int mode, float *out, double *out_f64 /* may be NULL */)
{
    const int OH = (H + 2 * padding - KH) / stride + 1; /* :210 */
    const int OW = (W + 2 * padding - KW) / stride + 1; /* :211 */
    if (OH <= 0 || OW <= 0) return QE_ORACLE_OK;
    const int64_t total = (int64_t)N * OC * OH * OW;

#pragma omp parallel for schedule(static)
    for (int64_t index = 0; index < total; index++) {
        int64_t batch = index / ((int64_t)OC * OH * OW);          /* :83 */
        int64_t output_index = index % ((int64_t)OC * OH * OW);   /* :86 */
        int outw = (int)(output_index % OW);                      /* :87 */
        int outh = (int)((output_index / OW) % OH);               /* :88 */
        int outc = (int)((output_index / ((int64_t)OH * OW)) % OC); /* :89 */

        float acc = bias ? bias[outc] : 0.0f;                     /* :92 */
        double acc64 = bias ? (double)bias[outc] : 0.0;

        for (int inc = 0; inc < IC; inc++)
            for (int keh = 0; keh < KH; keh++)
                for (int kew = 0; kew < KW; kew++) {
                    int inh = outh * stride + keh - padding;      /* :98 */
                    int inw = outw * stride + kew - padding;      /* :99 */
                    if (inh >= 0 && inh < H && inw >= 0 && inw < W) { /* :101 */
                        int64_t xi = batch * IC * H * W + (int64_t)inc * H * W + (int64_t)inh * W + inw; /* :103 */
                        int qx = qe_unpack_elem(x, xi, x_bits, x_sign);
                        float zx = x_per_tensor ? x_zero[0] : x_zero[inc];
                        float sx = x_per_tensor ? x_scale[0] : x_scale[inc];
                        int64_t wi = (int64_t)outc * IC * KH * KW + (int64_t)inc * KH * KW + keh * KW + kew; /* :118 */
                        int qw = qe_unpack_elem(w, wi, w_bits, w_sign);
                        float zw = w_per_tensor ? w_zero[0] : w_zero[outc];
                        float sw = w_per_tensor ? w_scale[0] : w_scale[outc];
                        if (mode == 2) {
                            acc64 += (((double)qx - (double)zx) * (double)sx) *
                                     (((double)qw - (double)zw) * (double)sw);
                        } else {
                            float xf = ((float)qx - zx) * sx;     /* :113-115 */
                            float wf = ((float)qw - zw) * sw;     /* :128-130 */
                            if (mode == 1) acc = fmaf(xf, wf, acc);
                            else           acc += xf * wf;        /* :133 */
                        }
                    }
                }
        if (mode == 2) {
            out[index] = (float)acc64;
            if (out_f64) out_f64[index] = acc64;
        } else {
            out[index] = acc;                                     /* :140 */
        }
    }
    return QE_ORACLE_OK;
}

/* ------------------------------------------------------------------------- */
/* quantconv2d_float_input:                                                   */
/* engine/kernels/functions/quantconv2d_float_input.cu:69-120, shapes :162-179*/
/* Same modes as above.                                                       */
/* ------------------------------------------------------------------------- */
int qe_oracle_quantconv2d_float_input(
    const float *x,
    const uint8_t *w, int w_bits, int w_sign,
    const float *w_scale, const float *w_zero, int w_per_tensor,
    const float *bias,
    int N, int IC, int H, int W, int OC, int KH, int KW, int stride, int padding,
    int mode, float *out, double *out_f64)
{
    const int OH = (H + 2 * padding - KH) / stride + 1; /* :177 */
    const int OW = (W + 2 * padding - KW) / stride + 1; /* :178 */
    if (OH <= 0 || OW <= 0) return QE_ORACLE_OK;
    const int64_t total = (int64_t)N * OC * OH * OW;

#pragma omp parallel for schedule(static)
    for (int64_t index = 0; index < total; index++) {
        int64_t batch = index / ((int64_t)OC * OH * OW);          /* :74 */
        int64_t output_index = index % ((int64_t)OC * OH * OW);   /* :77 */
        int outw = (int)(output_index % OW);
        int outh = (int)((output_index / OW) % OH);
        int outc = (int)((output_index / ((int64_t)OW * OH)) % OC);

        float acc = bias ? bias[outc] : 0.0f;                     /* :83 */
        double acc64 = bias ? (double)bias[outc] : 0.0;

        for (int inc = 0; inc < IC; inc++)
            for (int keh = 0; keh < KH; keh++)
                for (int kew = 0; kew < KW; kew++) {
                    int inh = outh * stride - padding + keh;      /* :89 */
                    int inw = outw * stride - padding + kew;      /* :90 */
                    if (inh >= 0 && inh < H && inw >= 0 && inw < W) { /* :92 */
                        int64_t wi = (int64_t)outc * IC * KH * KW + (int64_t)inc * KH * KW + keh * KW + kew; /* :94 */
                        int qw = qe_unpack_elem(w, wi, w_bits, w_sign);
                        float zw = w_per_tensor ? w_zero[0] : w_zero[outc];
                        float sw = w_per_tensor ? w_scale[0] : w_scale[outc];
                        float xf = x[batch * IC * H * W + (int64_t)inc * H * W + (int64_t)inh * W + inw]; /* :109 */
                        if (mode == 2) {
                            acc64 += (double)xf * (((double)qw - (double)zw) * (double)sw);
                        } else {
                            float wf = ((float)qw - zw) * sw;     /* :104-106 */
                            if (mode == 1) acc = fmaf(xf, wf, acc);
                            else           acc += xf * wf;        /* :112 */
                        }
                    }
                }
        if (mode == 2) {
            out[index] = (float)acc64;
            if (out_f64) out_f64[index] = acc64;
        } else {
            out[index] = acc;                                     /* :116 */
        }
    }
    return QE_ORACLE_OK;
}

/* ------------------------------------------------------------------------- */
/* quantlinear: engine/kernels/functions/quantlinear.cu:39-133 (one output     */
/* element per thread, 32-wide K tiles) restated as the per-element sum.       */
/* NOTE the conventions of THIS kernel, which differ from the conv kernels:    */
/*   * (q + zero): input_value = q_x + input_zero[row]   (:113-115)            */
/*                 weight_value = q_w + weight_zero[col] (:118-120)            */
/*   * the input scale/zero are indexed by the batch ROW, the weight's by the  */
/*     output COLUMN; s_scale = input_scale[row] * weight_scale[col] is one    */
/*     fp32 product (:96) applied inside the loop: tmp += x * w * s (:123)     */
/*   * tmp starts at 0 and the bias is added last (:131)                       */
/* x_per_tensor / w_per_tensor: the host expands 0-dim scales (:276-290); this */
/* restatement broadcasts element 0 instead.                                   */
/* The reference reads stale shared memory for K % 32 != 0 (no zero fill,      */
/* :76-92); the restatement is the mathematically intended sum over k < K.     */
/* modes as in qe_oracle_quantconv2d.                                          */
/* ------------------------------------------------------------------------- */
int qe_oracle_quantlinear(
    const uint8_t *x, int x_bits, int x_sign,
    const float *x_scale, const float *x_zero, int x_per_tensor,
    const uint8_t *w, int w_bits, int w_sign,
    const float *w_scale, const float *w_zero, int w_per_tensor,
    const float *bias, /* may be NULL (host substitutes zeros, :268) */
    int B, int K, int O, int mode, float *out, double *out_f64)
{
    const int64_t total = (int64_t)B * O;
#pragma omp parallel for schedule(static)
    for (int64_t index = 0; index < total; index++) {
        const int row = (int)(index / O), col = (int)(index % O);
        const float zx = x_per_tensor ? x_zero[0] : x_zero[row];
        const float zw = w_per_tensor ? w_zero[0] : w_zero[col];
        const float sx = x_per_tensor ? x_scale[0] : x_scale[row];
        const float sw = w_per_tensor ? w_scale[0] : w_scale[col];
        const float s = sx * sw;                                   /* :96 */
        float tmp = 0.0f;                                          /* :70 */
        double tmp64 = 0.0;
        for (int k = 0; k < K; k++) {
            int qx = qe_unpack_elem(x, (int64_t)row * K + k, x_bits, x_sign);   /* :78-83 */
            int qw = qe_unpack_elem(w, (int64_t)col * K + k, w_bits, w_sign);   /* :87-92 */
            if (mode == 2) {
                tmp64 += ((double)qx + (double)zx) * ((double)qw + (double)zw) * ((double)sx * (double)sw);
            } else {
                float iv = (float)qx + zx;                         /* :115 */
                float wv = (float)qw + zw;                         /* :120 */
                if (mode == 1) tmp = fmaf(iv * wv, s, tmp);
                else           tmp += iv * wv * s;                 /* :123 */
            }
        }
        if (mode == 2) {
            double v = tmp64 + (bias ? (double)bias[col] : 0.0);
            out[index] = (float)v;
            if (out_f64) out_f64[index] = v;
        } else {
            out[index] = tmp + (bias ? bias[col] : 0.0f);          /* :131 */
        }
    }
    return QE_ORACLE_OK;
}

/* ------------------------------------------------------------------------- */
/* quantlinear_float_input: functions/quantlinear_float_input.cu:36-104.      */
/* (q - zero) * scale convention for the weight (:82-86), fp32 input, sum from */
/* 0 and the bias added last (:102).  Same K-tail remark and modes as above.   */
/* ------------------------------------------------------------------------- */
int qe_oracle_quantlinear_float_input(
    const float *x,
    const uint8_t *w, int w_bits, int w_sign,
    const float *w_scale, const float *w_zero, int w_per_tensor,
    const float *bias,
    int B, int K, int O, int mode, float *out, double *out_f64)
{
    const int64_t total = (int64_t)B * O;
#pragma omp parallel for schedule(static)
    for (int64_t index = 0; index < total; index++) {
        const int row = (int)(index / O), col = (int)(index % O);
        const float zw = w_per_tensor ? w_zero[0] : w_zero[col];
        const float sw = w_per_tensor ? w_scale[0] : w_scale[col];
        float acc = 0.0f;                                          /* :62 */
        double acc64 = 0.0;
        for (int k = 0; k < K; k++) {
            int qw = qe_unpack_elem(w, (int64_t)col * K + k, w_bits, w_sign);   /* :73-77 */
            float xf = x[(int64_t)row * K + k];                    /* :68 */
            if (mode == 2) {
                acc64 += (double)xf * (((double)qw - (double)zw) * (double)sw);
            } else {
                float wf = ((float)qw - zw) * sw;                  /* :82-86 */
                if (mode == 1) acc = fmaf(xf, wf, acc);
                else           acc += xf * wf;                     /* :95 */
            }
        }
        if (mode == 2) {
            double v = acc64 + (bias ? (double)bias[col] : 0.0);
            out[index] = (float)v;
            if (out_f64) out_f64[index] = v;
        } else {
            out[index] = acc + (bias ? bias[col] : 0.0f);          /* :102 */
        }
    }
    return QE_ORACLE_OK;
}

/* Number of OpenMP threads the conv loops will use (for bench.py's `cores`). */
#ifdef _OPENMP
#include <omp.h>
int qe_oracle_num_threads(void) { return omp_get_max_threads(); }
void qe_oracle_set_num_threads(int n) { omp_set_num_threads(n); }
#else
int qe_oracle_num_threads(void) { return 1; }
void qe_oracle_set_num_threads(int n) { (void)n; }
#endif
