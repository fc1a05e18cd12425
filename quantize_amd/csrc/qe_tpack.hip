// qe_tpack.hip -- bit-packing / unpacking kernels for gfx950 (MI355X).
//
// Replaces engine/kernels/tpack/tpack.cu of the reference (tpack_cuda_kernel :30-84,
// tunpack_cuda_kernel :267-315).  Same bit stream (element i at bits [i*b,(i+1)*b),
// LSB first, value = q + (sign ? 2^(b-1) : 0)), different machine mapping:
//
//   reference: one thread = 8 elements, byte-granular read-modify-write of a
//              pre-zeroed output, lane stride 32 B on loads and b B on stores.
//   here     : one 256-thread workgroup = one tile of 8192 elements.
//              (1) coalesced vector loads (4 elements per lane per instruction,
//                  16 B/lane for fp32) -> 8-bit codes -> LDS (8 KiB);
//              (2) each thread turns 32 consecutive codes into exactly b 32-bit
//                  words in registers (32*b bits = b words: no word is shared
//                  between threads, so no RMW, no atomics, no pre-zeroing);
//              (3) the tile's 256*b words go back through LDS and leave as
//                  coalesced 16 B/lane stores.
//              The reference's range check (two full reductions + two host
//              syncs, tpack.cu:211-215) is fused into pass (1) as one flag.
//
// HBM-bound: algorithmic bytes per element = sizeof(T) + b/8 (pack),
// b/8 + 1 (unpack).  LDS traffic is 2 B/element, far below the LDS roof.
#include "qe_common.h"

namespace qe {

constexpr int TP_THREADS = 256;
constexpr int TP_EPT = 32;                    // elements per thread in the pack step
constexpr int TP_TILE = TP_THREADS * TP_EPT;  // 8192 elements per tile
constexpr int TP_MAX_BLOCKS = kNumCU * 8;

template <typename T>
struct alignas((sizeof(T) * 4 > 16) ? 16 : sizeof(T) * 4) Vec4 {
    T v[4];
};

// (char)x of tpack.cu:50 for in-range values, plus the range test of tpack.cu:211-215
// evaluated on the value as float (x.min().item<float>()).
template <typename T>
__device__ __forceinline__ unsigned tp_code(T v, float lo, float hi, unsigned offset, unsigned mask, bool &bad)
{
    const float f = (float)v;
    bad |= !(f >= lo && f <= hi);  // NaN fails both comparisons, like TORCH_CHECK
    const int iv = (int)v;         // truncation toward zero == (char)v while in range
    return ((unsigned)iv + offset) & mask;
}

// Fused activation quantisation: the Quantizer's
//   q = round(x / scale - zero).clamp(qmin, qmax)      (modelzoo/modules/quantizer.py:31, :215; zero in the MODULE's
// convention, i.e. subtracted here and added back on dequantisation) in front of the packer, so the fp32 integer-valued
// tensor the reference materialises between Quantizer and tpack (4 B/element written + read again) never exists.
// Same fp32 operations in the same order as torch: IEEE division, subtraction, round-half-even, clamp (NaN passes
// through the clamp and trips the range flag like it trips CHECK_RANGE).
struct TpQuant {
    const float *scale, *zero;   // 1 element, or one per channel
    float qmin, qmax;
    uint32_t inner, n_ch;        // per channel: channel of element i = (i / inner) % n_ch
};
__device__ __forceinline__ float tp_quantize(float v, float sc, float zr, float qmin, float qmax)
{
    const float r = rintf(v / sc - zr);
    return (r != r) ? r : fminf(fmaxf(r, qmin), qmax);
}

// QM: 0 = plain tpack, 1 = quantise with per-tensor scale/zero, 2 = per channel (T = float for 1 and 2)
template <typename T, int B, int QM>
__global__ __launch_bounds__(TP_THREADS) void tpack_kernel(
    const T *__restrict__ x, uint8_t *__restrict__ out, int64_t n, int64_t n_out_bytes,
    unsigned offset, float lo, float hi, int32_t *__restrict__ status, int in_aligned, int out_aligned, const TpQuant q)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_codes[TP_TILE / 4];    // 8 KiB
    __shared__ __attribute__((aligned(16))) uint32_t s_words[TP_THREADS * B]; // <= 8 KiB

    const int tid = threadIdx.x;
    const int64_t n_tiles = (n + TP_TILE - 1) / TP_TILE;
    constexpr unsigned mask = (1u << B) - 1u;
    bool bad = false;

    float sc1 = 1.0f, zr1 = 0.0f;
    if constexpr (QM == 1) { sc1 = q.scale[0]; zr1 = q.zero[0]; }
    // element -> code: the Quantizer in front of the packer when QM != 0 (ch = channel of the element)
    auto code = [&](T v, uint32_t ch) __attribute__((always_inline)) -> unsigned {
        if constexpr (QM == 0) return tp_code<T>(v, lo, hi, offset, mask, bad);
        else if constexpr (QM == 1) return tp_code<float>(tp_quantize((float)v, sc1, zr1, q.qmin, q.qmax), lo, hi, offset, mask, bad);
        else return tp_code<float>(tp_quantize((float)v, q.scale[ch], q.zero[ch], q.qmin, q.qmax), lo, hi, offset, mask, bad);
    };

    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t base = tile * TP_TILE;
        // per channel: plane and offset inside it of the tile's first element, once per tile in 64 bits; 32 bits from there
        uint32_t pl0 = 0, r0 = 0;
        if constexpr (QM == 2) {
            const int64_t pl = base / (int64_t)q.inner;
            r0 = (uint32_t)(base - pl * (int64_t)q.inner);
            pl0 = (uint32_t)(pl % (int64_t)q.n_ch);
        }
        auto chan = [&](uint32_t o) __attribute__((always_inline)) -> uint32_t {   // channel of element base + o
            if constexpr (QM == 2) return (pl0 + (r0 + o) / q.inner) % q.n_ch; else return 0u;
        };

        // (1) coalesced loads, 16 bytes per lane where the element type allows: a lane takes EPL consecutive elements
        // (16 for 1-byte inputs, 8 for 2-byte, 4 otherwise); iteration k covers elements base + k*256*EPL + EPL*tid ..
        // (with 4 one-byte elements per lane, int8 / uint8 inputs packed at 2.0-2.2 TB/s: 4-byte loads)
        constexpr int EPL = sizeof(T) == 1 ? 16 : (sizeof(T) == 2 ? 8 : 4);
#pragma unroll
        for (int k = 0; k < TP_EPT / EPL; ++k) {
            const uint32_t o0 = (uint32_t)(k * (TP_THREADS * EPL) + tid * EPL);
#pragma unroll
            for (int v4 = 0; v4 < EPL / 4; ++v4) {
                const uint32_t o = o0 + 4 * v4;
                const int64_t e = base + o;
                unsigned c0 = 0, c1 = 0, c2 = 0, c3 = 0;
                if (e + 3 < n && in_aligned && (QM != 2 || (q.inner & 3u) == 0)) {   // 4 elements of one channel
                    const Vec4<T> v = *reinterpret_cast<const Vec4<T> *>(x + e);      // adjacent Vec4 loads merge into one 16-byte load
                    const uint32_t ch = chan(o);
                    c0 = code(v.v[0], ch);
                    c1 = code(v.v[1], ch);
                    c2 = code(v.v[2], ch);
                    c3 = code(v.v[3], ch);
                } else {
                    if (e + 0 < n) c0 = code(x[e + 0], chan(o));
                    if (e + 1 < n) c1 = code(x[e + 1], chan(o + 1));
                    if (e + 2 < n) c2 = code(x[e + 2], chan(o + 2));
                    if (e + 3 < n) c3 = code(x[e + 3], chan(o + 3));
                }
                s_codes[o >> 2] = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
            }
        }
        __syncthreads();

        // (2) 32 consecutive codes -> B words, all in registers.
        {
            const uint4 lo4 = *reinterpret_cast<const uint4 *>(&s_codes[tid * 8]);
            const uint4 hi4 = *reinterpret_cast<const uint4 *>(&s_codes[tid * 8 + 4]);
            const uint32_t cw[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
            if constexpr (B == 8) {
#pragma unroll
                for (int i = 0; i < 8; ++i) s_words[tid * 8 + i] = cw[i];
            } else {
                uint64_t acc = 0;
                int nb = 0, wi = 0;
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    const uint32_t c = (cw[j >> 2] >> ((j & 3) * 8)) & 0xffu;
                    acc |= (uint64_t)c << nb;
                    nb += B;
                    if (nb >= 32) {
                        s_words[tid * B + wi] = (uint32_t)acc;
                        ++wi;
                        acc >>= 32;
                        nb -= 32;
                    }
                }
            }
        }
        __syncthreads();

        // (3) coalesced 16 B/lane stores of the tile's 256*B words.
        {
            const int64_t tile_byte0 = tile * (int64_t)(TP_TILE / 8) * B;
            const int64_t remain = n_out_bytes - tile_byte0;
            const int valid = (int)(remain < (int64_t)(TP_TILE / 8) * B ? remain : (int64_t)(TP_TILE / 8) * B);
            uint8_t *o = out + tile_byte0;
            for (int i = tid; i < (TP_THREADS * B) / 4; i += TP_THREADS) {
                const int off = i * 16;
                if (off >= valid) break;
                const uint4 w4 = *reinterpret_cast<const uint4 *>(&s_words[i * 4]);
                if (off + 16 <= valid && out_aligned) {
                    *reinterpret_cast<uint4 *>(o + off) = w4;
                } else {
                    const uint32_t ww[4] = {w4.x, w4.y, w4.z, w4.w};
                    const int lim = (valid - off) < 16 ? (valid - off) : 16;
                    for (int b = 0; b < lim; ++b) o[off + b] = (uint8_t)(ww[b >> 2] >> ((b & 3) * 8));
                }
            }
        }
        __syncthreads();  // LDS is reused by the next tile
    }

    if (status != nullptr) {
        // one atomic per wave at most (the compiler folds the ballot)
        if (__any(bad)) {
            if ((threadIdx.x & (kWave - 1)) == 0) atomicOr(status, 1);
        }
    }
}

template <int B>
__global__ __launch_bounds__(TP_THREADS) void tunpack_kernel(
    const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int64_t n, int64_t n_in_bytes,
    unsigned offset, int in_aligned, int out_aligned)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_words[TP_THREADS * B];
    __shared__ __attribute__((aligned(16))) uint32_t s_codes[TP_TILE / 4];

    const int tid = threadIdx.x;
    const int64_t n_tiles = (n + TP_TILE - 1) / TP_TILE;
    constexpr unsigned mask = (1u << B) - 1u;

    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        // (1) coalesced load of the tile's packed words into LDS
        {
            const int64_t tile_byte0 = tile * (int64_t)(TP_TILE / 8) * B;
            const int64_t remain = n_in_bytes - tile_byte0;
            const int valid = (int)(remain < (int64_t)(TP_TILE / 8) * B ? remain : (int64_t)(TP_TILE / 8) * B);
            const uint8_t *p = in + tile_byte0;
            for (int i = tid; i < (TP_THREADS * B) / 4; i += TP_THREADS) {
                const int off = i * 16;
                uint4 w4 = make_uint4(0, 0, 0, 0);
                if (off + 16 <= valid && in_aligned) {
                    w4 = *reinterpret_cast<const uint4 *>(p + off);
                } else if (off < valid) {
                    uint32_t ww[4] = {0, 0, 0, 0};
                    const int lim = (valid - off) < 16 ? (valid - off) : 16;
                    for (int b = 0; b < lim; ++b) ww[b >> 2] |= (uint32_t)p[off + b] << ((b & 3) * 8);
                    w4 = make_uint4(ww[0], ww[1], ww[2], ww[3]);
                }
                *reinterpret_cast<uint4 *>(&s_words[i * 4]) = w4;
            }
        }
        __syncthreads();

        // (2) B words -> 32 codes -> minus offset (mod 256, tpack.cu:311) -> LDS bytes
        {
            uint32_t cw[8];
            if constexpr (B == 8) {
#pragma unroll
                for (int i = 0; i < 8; ++i) cw[i] = s_words[tid * 8 + i];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    // per-byte subtract of `offset` (0 or 128): x - 128 == x ^ 0x80 (mod 256)
                    cw[i] = offset ? (cw[i] ^ 0x80808080u) : cw[i];
                }
            } else {
                uint32_t w[B];
#pragma unroll
                for (int i = 0; i < B; ++i) w[i] = s_words[tid * B + i];
#pragma unroll
                for (int i = 0; i < 8; ++i) cw[i] = 0;
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    const int bit = j * B;
                    const int wi = bit >> 5, sh = bit & 31;
                    uint32_t c = w[wi] >> sh;
                    if (sh + B > 32) c |= w[wi + 1] << (32 - sh);
                    c = ((c & mask) - offset) & 0xffu;
                    cw[j >> 2] |= c << ((j & 3) * 8);
                }
            }
            *reinterpret_cast<uint4 *>(&s_codes[tid * 8]) = make_uint4(cw[0], cw[1], cw[2], cw[3]);
            *reinterpret_cast<uint4 *>(&s_codes[tid * 8 + 4]) = make_uint4(cw[4], cw[5], cw[6], cw[7]);
        }
        __syncthreads();

        // (3) coalesced 16 B/lane stores of 8192 bytes
        {
            const int64_t base = tile * TP_TILE;
            const int64_t remain = n - base;
            const int valid = (int)(remain < TP_TILE ? remain : TP_TILE);
            uint8_t *o = out + base;
#pragma unroll
            for (int k = 0; k < TP_TILE / 16 / TP_THREADS; ++k) {
                const int i = k * TP_THREADS + tid;
                const int off = i * 16;
                if (off >= valid) break;
                const uint4 c4 = *reinterpret_cast<const uint4 *>(&s_codes[i * 4]);
                if (off + 16 <= valid && out_aligned) {
                    *reinterpret_cast<uint4 *>(o + off) = c4;
                } else {
                    const uint32_t ww[4] = {c4.x, c4.y, c4.z, c4.w};
                    const int lim = (valid - off) < 16 ? (valid - off) : 16;
                    for (int b = 0; b < lim; ++b) o[off + b] = (uint8_t)(ww[b >> 2] >> ((b & 3) * 8));
                }
            }
        }
        __syncthreads();
    }
}

template <typename T, int QM = 0>
static int launch_tpack_t(const void *x, int64_t n, int n_bits, int sign, uint8_t *out, int32_t *status, hipStream_t s,
                          const TpQuant q = TpQuant{nullptr, nullptr, 0.0f, 0.0f, 1u, 1u})
{
    const int64_t n_out = qe_packed_nbytes(n, n_bits);
    const unsigned offset = sign ? (1u << (n_bits - 1)) : 0u;  // tpack.cu:108-111
    const float lo = sign ? -(float)(1 << (n_bits - 1)) : 0.0f;
    const float hi = sign ? (float)((1 << (n_bits - 1)) - 1) : (float)((1 << n_bits) - 1);
    const int64_t n_tiles = (n + TP_TILE - 1) / TP_TILE;
    const int blocks = (int)(n_tiles < TP_MAX_BLOCKS ? n_tiles : TP_MAX_BLOCKS);
    const size_t va = sizeof(Vec4<T>) > 16 ? 16 : sizeof(Vec4<T>);
    const int in_al = ((uintptr_t)x % va) == 0;
    const int out_al = ((uintptr_t)out % 16) == 0;
    const T *xp = static_cast<const T *>(x);
#define QE_TP_CASE(BITS)                                                                          \
    case BITS:                                                                                    \
        hipLaunchKernelGGL((tpack_kernel<T, BITS, QM>), dim3(blocks), dim3(TP_THREADS), 0, s, xp, out, \
                           n, n_out, offset, lo, hi, status, in_al, out_al, q);                   \
        break;
    switch (n_bits) {
        QE_TP_CASE(1) QE_TP_CASE(2) QE_TP_CASE(3) QE_TP_CASE(4)
        QE_TP_CASE(5) QE_TP_CASE(6) QE_TP_CASE(7) QE_TP_CASE(8)
        default: return QE_ERR_NBITS;
    }
#undef QE_TP_CASE
    QE_LAUNCH_CHECK();
    return QE_OK;
}

}  // namespace qe

extern "C" int64_t qe_packed_nbytes(int64_t n_elements, int n_bits)
{
    if (n_elements <= 0 || n_bits <= 0) return 0;
    return (n_elements * (int64_t)n_bits + 7) / 8;
}

extern "C" int qe_tpack(const void *x, int dtype, int64_t n, int n_bits, int sign,
                        uint8_t *out, int32_t *status, qe_stream_t stream)
{
    using namespace qe;
    if (!(n_bits > 0 && n_bits <= 8)) return QE_ERR_NBITS;
    if (n < 0) return QE_ERR_ARG;
    if (n == 0) return QE_OK;
    if (x == nullptr || out == nullptr) return QE_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case QE_U8:  return launch_tpack_t<uint8_t>(x, n, n_bits, sign, out, status, s);
        case QE_I8:  return launch_tpack_t<int8_t>(x, n, n_bits, sign, out, status, s);
        case QE_I16: return launch_tpack_t<int16_t>(x, n, n_bits, sign, out, status, s);
        case QE_I32: return launch_tpack_t<int32_t>(x, n, n_bits, sign, out, status, s);
        case QE_I64: return launch_tpack_t<int64_t>(x, n, n_bits, sign, out, status, s);
        case QE_F16: return launch_tpack_t<_Float16>(x, n, n_bits, sign, out, status, s);
        case QE_F32: return launch_tpack_t<float>(x, n, n_bits, sign, out, status, s);
        case QE_F64: return launch_tpack_t<double>(x, n, n_bits, sign, out, status, s);
        default: return QE_ERR_DTYPE;
    }
}

extern "C" int qe_quantize_pack(const float *x, int64_t n, const float *scale, const float *zero, int32_t n_param,
                                int64_t inner, float qmin, float qmax, int n_bits, int sign, uint8_t *out,
                                int32_t *status, qe_stream_t stream)
{
    using namespace qe;
    if (!(n_bits > 0 && n_bits <= 8)) return QE_ERR_NBITS;
    if (n < 0 || n_param < 1) return QE_ERR_ARG;
    if (n == 0) return QE_OK;
    if (x == nullptr || out == nullptr || scale == nullptr || zero == nullptr) return QE_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    TpQuant q{scale, zero, qmin, qmax, 1u, 1u};
    if (n_param == 1) return launch_tpack_t<float, 1>(x, n, n_bits, sign, out, status, s, q);
    if (inner < 1 || inner >= (1ll << 31) - TP_TILE) return QE_ERR_ARG;   // 32-bit offsets inside a tile
    q.inner = (uint32_t)inner;
    q.n_ch = (uint32_t)n_param;
    return launch_tpack_t<float, 2>(x, n, n_bits, sign, out, status, s, q);
}

namespace qe {
// Sub-8-bit packed stream -> one byte per element holding the SIGNED 8-bit stored code q + 128 (what tpack(q, 8, true)
// would have produced).  The conv front end uses it to run b < 8 activations on the 8-bit MFMA kernels.  Same kernel
// as tunpack: it computes (code - offset) mod 256, and q + 128 == code - offset + 128 == code - (offset + 128) mod 256.
int expand_codes_s8(const uint8_t *packed, int64_t n, int n_bits, int sign, uint8_t *out, hipStream_t s)
{
    if (!(n_bits > 0 && n_bits < 8)) return QE_ERR_NBITS;
    if (n <= 0) return QE_OK;
    const int64_t n_in = qe_packed_nbytes(n, n_bits);
    const unsigned offset = ((sign ? (1u << (n_bits - 1)) : 0u) + 128u) & 0xffu;
    const int64_t n_tiles = (n + TP_TILE - 1) / TP_TILE;
    const int blocks = (int)(n_tiles < TP_MAX_BLOCKS ? n_tiles : TP_MAX_BLOCKS);
    const int in_al = ((uintptr_t)packed % 16) == 0;
    const int out_al = ((uintptr_t)out % 16) == 0;
#define QE_EX_CASE(BITS)                                                                              \
    case BITS:                                                                                        \
        hipLaunchKernelGGL((tunpack_kernel<BITS>), dim3(blocks), dim3(TP_THREADS), 0, s, packed, out, \
                           n, n_in, offset, in_al, out_al);                                           \
        break;
    switch (n_bits) {
        QE_EX_CASE(1) QE_EX_CASE(2) QE_EX_CASE(3) QE_EX_CASE(4) QE_EX_CASE(5) QE_EX_CASE(6) QE_EX_CASE(7)
        default: return QE_ERR_NBITS;
    }
#undef QE_EX_CASE
    QE_LAUNCH_CHECK();
    return QE_OK;
}
}  // namespace qe

extern "C" int qe_tunpack(const uint8_t *packed, int64_t n, int n_bits, int sign,
                          void *out, qe_stream_t stream)
{
    using namespace qe;
    if (!(n_bits > 0 && n_bits <= 8)) return QE_ERR_NBITS;
    if (n < 0) return QE_ERR_ARG;
    if (n == 0) return QE_OK;
    if (packed == nullptr || out == nullptr) return QE_ERR_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t n_in = qe_packed_nbytes(n, n_bits);
    const unsigned offset = sign ? (1u << (n_bits - 1)) : 0u;  // tpack.cu:339-342
    const int64_t n_tiles = (n + TP_TILE - 1) / TP_TILE;
    const int blocks = (int)(n_tiles < TP_MAX_BLOCKS ? n_tiles : TP_MAX_BLOCKS);
    const int in_al = ((uintptr_t)packed % 16) == 0;
    const int out_al = ((uintptr_t)out % 16) == 0;
    uint8_t *o = static_cast<uint8_t *>(out);
#define QE_TU_CASE(BITS)                                                                            \
    case BITS:                                                                                      \
        hipLaunchKernelGGL((tunpack_kernel<BITS>), dim3(blocks), dim3(TP_THREADS), 0, s, packed, o, \
                           n, n_in, offset, in_al, out_al);                                         \
        break;
    switch (n_bits) {
        QE_TU_CASE(1) QE_TU_CASE(2) QE_TU_CASE(3) QE_TU_CASE(4)
        QE_TU_CASE(5) QE_TU_CASE(6) QE_TU_CASE(7) QE_TU_CASE(8)
        default: return QE_ERR_NBITS;
    }
#undef QE_TU_CASE
    QE_LAUNCH_CHECK();
    return QE_OK;
}
