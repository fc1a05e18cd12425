// qe_conv_mfma.hip -- int8 MFMA implicit-GEMM convolution for gfx950 (MI355X).
//
// Replaces quantconv2d_cuda_kernel (engine/kernels/functions/quantconv2d.cu:49-142) for
// every problem whose activation scale/zero is per tensor (the reference's default
// granularity).  The reference unpacks and dequantises both operands to fp32 inside the
// innermost loop of a one-thread-per-output kernel; here the integer part of the sum is
// done exactly on the matrix cores and the scales are applied once per output:
//
//   out[n,oc,p] = bias[oc] + sum_inb ((qx - zx) sx) ((qw - zw[oc]) sw[oc])
//               = bias[oc] + sx sw[oc] * ( S_aw - zw' S_x - zx' S_w + N_inb zx' zw' )
//   a_x = qx - d_x, a_w = qw - d_w   (d = 128 for unsigned 8-bit, else 0: fits int8)
//   zx' = zx - d_x, zw' = zw - d_w   (floats, may be non-integer: minmax.py:143)
//   S_aw = sum_inb a_x a_w           v_mfma_i32_32x32x32_i8, exact int32; padded taps hold a_x = 0
//   S_x  = sum_inb a_x               v_dot4 on the same B fragments (only if some zw' != 0)
//   S_w  = sum_inb a_w               per-(oc,tap) table from the prep pass, border aware
//                                    (only if zx' != 0); N_inb = IC * #in-bounds taps
//   "inb" = taps inside the image: the reference SKIPS padded taps (quantconv2d.cu:101).
//
// Two kernels per call:
//   prep  : packed OIHW weights (any 1..8 bits) -> int8 a_w in MFMA A-fragment order
//           Wt[tap][ic/16][oc][16], per-oc epilogue constants, per-(oc,tap) sums.
//   main  : GEMM view  D[oc, pixel] = sum_k Wt[oc,k] X[k,pixel],  k = (ic-chunk, tap, ic%32).
//           Workgroup = 256 threads = 4 waves; tile = MT output channels x (TH output rows x
//           full width) pixels of ONE image (<= 256 pixels = 8 MFMA column tiles).  D has the
//           pixel on the lane (32x32 C/D map: col = lane&31), so every accumulator register
//           stores as two 128-byte row segments of the fp32 NCHW output: no epilogue transpose.
//           Activations: NCHW bytes have K strided by H*W, MFMA wants 16 K-contiguous bytes per
//           lane.  Each staging thread loads 16 channels x 4 pixels (16 dwords, coalesced along
//           the row), transposes 4x4 byte blocks with v_perm_b32 and writes one 16-byte
//           [pixel][16 ch] vector per pixel into LDS.  The LDS image is the input halo tile in
//           pixel-major order with zeroed borders: a tap is just a constant LDS offset, so there
//           is no im2col expansion and no per-tap bounds test in the inner loop.
//           Weights: A fragments go L2 -> VGPR directly (each wave owns a distinct 32-row strip,
//           LDS would add a copy without any sharing); all taps of a chunk are requested before
//           the next chunk's activation loads so the in-order vmcnt never parks a fast L2 hit
//           behind an HBM miss.
//           Pipeline per 32-channel chunk: request A(c) -> transpose X(c) regs into LDS -> barrier
//           -> request X(c+1) into registers -> MFMA over all taps -> barrier.
#include "qe_common.h"

#include <algorithm>

namespace qe {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int MF_THREADS = 256;
constexpr int MF_UNITS = 2;        // staging units (16 ch x 4 px) per thread per chunk
constexpr int MF_MAX_NTILES = 8;   // 32-pixel column tiles per workgroup (7 in the 4x1 wave layout)
constexpr int MF_MAX_LDS = 64 * 1024;

struct MfmaArgs {
    const uint8_t *x;
    int64_t x_bytes;           // length of the packed activation stream
    const float *x_zero;       // per tensor
    int x_bits, x_sign;
    const int8_t *wt;          // [KK][NG][OCP][16]
    const float *ep;           // [3][OCP]: alpha = sx*sw, zw', bias
    const int *ws;             // [OCP][KK+1]: sum_ic a_w per tap, [KK] = all taps
    float *out;
    int N, IC, H, W, OC, KH, KW, stride, pad, OH, OW;
    int OCP, NG, NCH;          // padded oc, 16-channel groups (even), 32-channel chunks
    int TH, tiles_h, n_pix_tiles, n_oc_tiles;
    int IHT, IWP, ROWMUL, COLMUL, ni;
};

struct PrepArgs {
    const uint8_t *w;
    const float *w_scale, *w_zero;
    const float *x_scale;
    const float *bias;
    int w_bits, w_sign, w_per_tensor;
    int OC, IC, KK, OCP, NG;
    int8_t *wt;
    float *ep;
    int *ws;
};

__device__ __forceinline__ int unpack_code(const uint8_t *__restrict__ p, int64_t ele_idx, int n_bits)
{
    const int64_t bit = ele_idx * n_bits;
    const int64_t byte_idx = bit >> 3;
    const int bit_idx = (int)(bit & 7);
    unsigned v = ((unsigned)p[byte_idx] >> bit_idx);
    if (bit_idx + n_bits > 8) v |= ((unsigned)p[byte_idx + 1] << (8 - bit_idx));
    return (int)(v & ((1u << n_bits) - 1u));
}

// stored code u -> MFMA operand a = q - d = u - c, c = off (signed) | 128 (unsigned 8-bit) | 0
__host__ __device__ __forceinline__ int code_bias(int n_bits, int sign)
{
    return sign ? (1 << (n_bits - 1)) : (n_bits == 8 ? 128 : 0);
}
// d: what was subtracted from q on top of the sign offset (added back through the zero point)
__host__ __device__ __forceinline__ float zero_shift(int n_bits, int sign)
{
    return (!sign && n_bits == 8) ? 128.0f : 0.0f;
}

// ---------------------------------------------------------------------------------------------
// prep: one workgroup per (padded) output channel.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_mfma_prep_kernel(const PrepArgs a)
{
    __shared__ int s_ws[64];
    const int oc = blockIdx.x;
    const int tid = threadIdx.x;
    if (tid < 64) s_ws[tid] = 0;
    __syncthreads();
    const int cb = code_bias(a.w_bits, a.w_sign);
    const bool live = oc < a.OC;
    for (int idx = tid; idx < a.KK * a.NG; idx += 256) {
        const int tap = idx / a.NG, icg = idx - tap * a.NG;
        uint32_t v[4] = {0, 0, 0, 0};
        int sum = 0;
        if (live) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int ic = icg * 16 + j;
                int aw = 0;
                if (ic < a.IC) {
                    const int64_t e = ((int64_t)oc * a.IC + ic) * a.KK + tap;  // quantconv2d.cu:118
                    aw = unpack_code(a.w, e, a.w_bits) - cb;
                }
                sum += aw;
                v[j >> 2] |= ((uint32_t)aw & 0xffu) << ((j & 3) * 8);
            }
        }
        *reinterpret_cast<uint4 *>(a.wt + (((int64_t)tap * a.NG + icg) * a.OCP + oc) * 16) =
            make_uint4(v[0], v[1], v[2], v[3]);
        if (sum != 0) atomicAdd(&s_ws[tap], sum);
    }
    __syncthreads();
    if (tid < a.KK) a.ws[oc * (a.KK + 1) + tid] = s_ws[tid];
    if (tid == 64) {
        int all = 0;
        for (int t = 0; t < a.KK; ++t) all += s_ws[t];
        a.ws[oc * (a.KK + 1) + a.KK] = all;
    }
    if (tid == 65) {
        float alpha = 0.0f, zwp = 0.0f, b = 0.0f;
        if (live) {
            const float sw = a.w_per_tensor ? a.w_scale[0] : a.w_scale[oc];
            const float zw = a.w_per_tensor ? a.w_zero[0] : a.w_zero[oc];
            alpha = a.x_scale[0] * sw;
            zwp = zw - zero_shift(a.w_bits, a.w_sign);
            b = a.bias ? a.bias[oc] : 0.0f;
        }
        a.ep[oc] = alpha;
        a.ep[a.OCP + oc] = zwp;
        a.ep[2 * a.OCP + oc] = b;
    }
}

// ---------------------------------------------------------------------------------------------
// activation fetch: 4 consecutive elements of one channel row -> 4 int8 operands in a dword
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fetch_quad8(const uint8_t *__restrict__ x, int64_t e, int64_t x_bytes)
{
    uint32_t v;
    if (e + 4 <= x_bytes) {
        __builtin_memcpy(&v, x + e, 4);  // one (possibly unaligned) global_load_dword
    } else {
        v = 0;
        for (int j = 0; j < 4; ++j)
            if (e + j < x_bytes) v |= (uint32_t)x[e + j] << (8 * j);
    }
    return v ^ 0x80808080u;  // u - 128 in every byte: signed q, or unsigned q - 128
}

__device__ __forceinline__ uint32_t fetch_quad_sub8(const uint8_t *__restrict__ x, int64_t e, int64_t x_bytes,
                                                   int n_bits, int cb)
{
    const int64_t bit = e * n_bits;
    const int64_t byte = bit >> 3;
    const int sh = (int)(bit & 7);
    uint64_t v;
    if (byte + 8 <= x_bytes) {
        __builtin_memcpy(&v, x + byte, 8);
    } else {
        v = 0;
        for (int j = 0; j < 8; ++j)
            if (byte + j < x_bytes) v |= (uint64_t)x[byte + j] << (8 * j);
    }
    v >>= sh;
    const uint32_t mask = (1u << n_bits) - 1u;
    uint32_t r = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t c = (uint32_t)(v >> (j * n_bits)) & mask;
        r |= ((c - (uint32_t)cb) & 0xffu) << (8 * j);
    }
    return r;
}

// 4x4 byte transpose: in d0..d3 (one channel each, 4 pixels), out o0..o3 (one pixel each, 4 channels)
__device__ __forceinline__ void transpose4x4(uint32_t d0, uint32_t d1, uint32_t d2, uint32_t d3,
                                             uint32_t &o0, uint32_t &o1, uint32_t &o2, uint32_t &o3)
{
    const uint32_t t0 = __builtin_amdgcn_perm(d1, d0, 0x05010400u);  // d0.b0 d1.b0 d0.b1 d1.b1
    const uint32_t t1 = __builtin_amdgcn_perm(d1, d0, 0x07030602u);  // d0.b2 d1.b2 d0.b3 d1.b3
    const uint32_t t2 = __builtin_amdgcn_perm(d3, d2, 0x05010400u);
    const uint32_t t3 = __builtin_amdgcn_perm(d3, d2, 0x07030602u);
    o0 = __builtin_amdgcn_perm(t2, t0, 0x05040100u);
    o1 = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
    o2 = __builtin_amdgcn_perm(t3, t1, 0x05040100u);
    o3 = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
}

// ---------------------------------------------------------------------------------------------
// main kernel.  WM x WN waves over (oc, pixel tiles); NIW column tiles per wave; KKT = taps known
// at compile time (1, 9) or 0 for a runtime tap loop.
// ---------------------------------------------------------------------------------------------
template <int WM, int WN, int NIW, int KKT>
__global__ __launch_bounds__(MF_THREADS, 2) void conv_mfma_kernel(const MfmaArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint4 *Xs = reinterpret_cast<uint4 *>(smem);

    constexpr int MT = 32 * WM;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int col = lane & 31, h = lane >> 5;

    // XCD-aware block map: blocks b and b+8 share an XCD (and its L2); give the oc-tiles of one
    // pixel tile ids that differ by multiples of 8 so they read the same activations from one L2.
    const int bid = blockIdx.x;
    const int grp_sz = 8 * a.n_oc_tiles;
    const int grp = bid / grp_sz, rem = bid - grp * grp_sz;
    const int pt = grp * 8 + (rem & 7);
    const int ot = rem >> 3;
    if (pt >= a.n_pix_tiles) return;

    const int n = pt / a.tiles_h;
    const int oh0 = (pt - n * a.tiles_h) * a.TH;
    const int th = min(a.TH, a.OH - oh0);
    const int NT = th * a.OW;
    const int ih0 = oh0 * a.stride - a.pad;
    const int GSZ = a.IHT * a.IWP;
    const int KK = (KKT > 0) ? KKT : a.KH * a.KW;

    // ---- zero the LDS halo image once: borders / padded channels are never written again ----
    for (int i = tid; i < 2 * GSZ; i += MF_THREADS) Xs[i] = make_uint4(0, 0, 0, 0);

    // ---- per-lane pixel bases of the wave's column tiles (uint4 index into Xs) ---------------
    const int RS = a.stride / a.ROWMUL, CS = a.stride / a.COLMUL;
    int pixidx[NIW];
    int niw = 0;  // column tiles this wave really has
#pragma unroll
    for (int t = 0; t < NIW; ++t) {
        const int tt = wn + t * WN;
        if (tt < a.ni) niw = t + 1;
        const int q = tt * 32 + col;
        int idx = h * GSZ;
        if (q < NT) {
            const int r = q / a.OW, c = q - r * a.OW;
            idx += (r * RS) * a.IWP + c * CS;
        }
        pixidx[t] = idx;
    }

    // ---- staging units of this thread (chunk invariant) -------------------------------------
    const int NQ = (a.W + 3) >> 2;
    const int units = 2 * a.IHT * NQ;
    const int64_t HW = (int64_t)a.H * a.W;
    int64_t u_off[MF_UNITS];
    int u_g[MF_UNITS];
    bool u_ok[MF_UNITS];
    int u_lds[MF_UNITS][4];
#pragma unroll
    for (int u = 0; u < MF_UNITS; ++u) {
        const int uid = tid + u * MF_THREADS;
        const int g = uid / (a.IHT * NQ);
        const int r = uid - g * (a.IHT * NQ);
        const int l = r / NQ, iq = r - l * NQ;
        const int ih = ih0 + l * a.ROWMUL;
        const bool ok = uid < units && ih >= 0 && ih < a.H;
        u_ok[u] = ok;
        u_g[u] = g;
        u_off[u] = (((int64_t)n * a.IC + g * 16) * a.H + ih) * a.W + 4 * iq;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int iw = 4 * iq + j;
            const int cl = iw + a.pad;
            const int clc = cl / a.COLMUL;
            const bool pok = ok && iw < a.W && (clc * a.COLMUL == cl) && clc < a.IWP;
            u_lds[u][j] = pok ? (g * a.IHT + l) * a.IWP + clc : -1;
        }
    }

    // ---- weight fragment pointer: lane (row col, half h) reads Wt[tap][2c + h][oc][16 B] ------
    const int8_t *a_ptr = a.wt + ((int64_t)h * a.OCP + ot * MT + wm * 32 + col) * 16;
    const int64_t grp_stride = (int64_t)a.OCP * 16;            // one 16-channel group
    const int64_t tap_stride = (int64_t)a.NG * grp_stride;     // one tap
    const int cbx = code_bias(a.x_bits, a.x_sign);

    v16i acc[NIW];
    int sxacc[NIW];
#pragma unroll
    for (int t = 0; t < NIW; ++t) {
        sxacc[t] = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0;
    }

    // does any output channel of this tile have zw' != 0 ?  (workgroup-uniform)
    int zw_local = 0;
    if (tid < MT) zw_local = (a.ep[a.OCP + ot * MT + tid] != 0.0f) ? 1 : 0;
    const bool need_sx = __syncthreads_or(zw_local) != 0;  // also orders the LDS zero fill

    uint32_t d[MF_UNITS][16];
    auto issue_x = [&](int c) {
#pragma unroll
        for (int u = 0; u < MF_UNITS; ++u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ic = c * 32 + u_g[u] * 16 + i;
                uint32_t v = 0;
                if (u_ok[u] && ic < a.IC) {
                    const int64_t e = u_off[u] + ((int64_t)c * 32 + i) * HW;
                    v = (a.x_bits == 8) ? fetch_quad8(a.x, e, a.x_bytes)
                                        : fetch_quad_sub8(a.x, e, a.x_bytes, a.x_bits, cbx);
                }
                d[u][i] = v;
            }
        }
    };

    issue_x(0);

    for (int c = 0; c < a.NCH; ++c) {
        // (1) request this chunk's weight fragments (L2 hits) ahead of everything else
        v4i afr[(KKT > 0) ? KKT : 1];
        const int8_t *a_c = a_ptr + (int64_t)(2 * c) * grp_stride;
        if constexpr (KKT > 0) {
#pragma unroll
            for (int tap = 0; tap < KKT; ++tap)
                afr[tap] = *reinterpret_cast<const v4i *>(a_c + tap * tap_stride);
        }

        // (2) transpose the prefetched activations into the LDS halo image
#pragma unroll
        for (int u = 0; u < MF_UNITS; ++u) {
            if (u_ok[u]) {
                uint32_t o[4][4];
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    transpose4x4(d[u][4 * m], d[u][4 * m + 1], d[u][4 * m + 2], d[u][4 * m + 3],
                                 o[0][m], o[1][m], o[2][m], o[3][m]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (u_lds[u][j] >= 0) Xs[u_lds[u][j]] = make_uint4(o[j][0], o[j][1], o[j][2], o[j][3]);
            }
        }
        __syncthreads();

        // (3) next chunk's activations go in flight under the MFMA phase
        if (c + 1 < a.NCH) issue_x(c + 1);

        // (4) all taps of this chunk
        if constexpr (KKT > 0) {
#pragma unroll
            for (int tap = 0; tap < KKT; ++tap) {
                const int kh = tap / ((KKT == 9) ? 3 : 1), kw = tap - kh * ((KKT == 9) ? 3 : 1);
                const int tapoff = kh * a.IWP + kw;
#pragma unroll
                for (int t = 0; t < NIW; ++t) {
                    if (t < niw) {
                        const v4i b = *reinterpret_cast<const v4i *>(&Xs[pixidx[t] + tapoff]);
                        acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(afr[tap], b, acc[t], 0, 0, 0);
                        if (need_sx) {
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                sxacc[t] = __builtin_amdgcn_sdot4(b[k], 0x01010101, sxacc[t], false);
                        }
                    }
                }
            }
        } else {
            for (int tap = 0; tap < KK; ++tap) {
                const int kh = tap / a.KW, kw = tap - kh * a.KW;
                const int tapoff = kh * a.IWP + kw;
                const v4i af = *reinterpret_cast<const v4i *>(a_c + tap * tap_stride);
#pragma unroll
                for (int t = 0; t < NIW; ++t) {
                    if (t < niw) {
                        const v4i b = *reinterpret_cast<const v4i *>(&Xs[pixidx[t] + tapoff]);
                        acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, b, acc[t], 0, 0, 0);
                        if (need_sx) {
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                sxacc[t] = __builtin_amdgcn_sdot4(b[k], 0x01010101, sxacc[t], false);
                        }
                    }
                }
            }
        }
        __syncthreads();  // everyone is done reading before the next chunk overwrites the image
    }

    // ---- epilogue -----------------------------------------------------------------------------
    const float zxp = a.x_zero[0] - zero_shift(a.x_bits, a.x_sign);
    const bool need_sw = zxp != 0.0f;
    const int oc_base = ot * MT + wm * 32 + 4 * h;  // + (reg&3) + 8*(reg>>2)
    float al[16], bi[16], zw[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int oc = oc_base + (r & 3) + 8 * (r >> 2);
        al[r] = a.ep[oc];
        zw[r] = a.ep[a.OCP + oc];
        bi[r] = a.ep[2 * a.OCP + oc];
    }
    const int64_t OHW = (int64_t)a.OH * a.OW;
    float *out_n = a.out + (int64_t)n * a.OC * OHW + (int64_t)oh0 * a.OW;

#pragma unroll
    for (int t = 0; t < NIW; ++t) {
        if (t < niw) {
            const int tt = wn + t * WN;
            const int q = tt * 32 + col;
            const bool valid = q < NT;
            int sxs = 0;
            if (need_sx) sxs = sxacc[t] + __shfl_xor(sxacc[t], 32);  // both 16-channel halves
            // in-bounds tap mask of this pixel (only needed for the zero-point terms)
            unsigned long long mask = 0;
            int n_inb = 0;
            bool interior = true;
            if (need_sw || need_sx) {
                const int r = valid ? q / a.OW : 0, c = valid ? q - r * a.OW : 0;
                const int ihb = (oh0 + r) * a.stride - a.pad, iwb = c * a.stride - a.pad;
                for (int tap = 0; tap < KK; ++tap) {
                    const int kh = tap / a.KW, kw = tap - kh * a.KW;
                    const bool inb = (ihb + kh) >= 0 && (ihb + kh) < a.H && (iwb + kw) >= 0 && (iwb + kw) < a.W;
                    if (inb) { mask |= 1ull << tap; ++n_inb; } else interior = false;
                }
            }
            const float fn = (float)(n_inb * a.IC);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int oc = oc_base + (r & 3) + 8 * (r >> 2);
                float v = (float)acc[t][r];
                if (need_sx) v = fmaf(-zw[r], (float)sxs, v);
                if (need_sw) {
                    int sw_sum;
                    const int *wsr = a.ws + (int64_t)oc * (KK + 1);
                    if (interior) {
                        sw_sum = wsr[KK];
                    } else {
                        sw_sum = 0;
                        for (int tap = 0; tap < KK; ++tap)
                            if ((mask >> tap) & 1ull) sw_sum += wsr[tap];
                    }
                    v = fmaf(-zxp, (float)sw_sum, v);
                    v = fmaf(fn * zxp, zw[r], v);
                }
                const float res = fmaf(al[r], v, bi[r]);
                if (valid && oc < a.OC) out_n[(int64_t)oc * OHW + q] = res;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct MfmaPlan {
    bool ok = false;
    int cfg = 0;       // 0: 4x1 waves (MT 128), 1: 2x2 (MT 64), 2: 1x4 (MT 32)
    int MT = 0, OCP = 0, NCH = 0, NG = 0, KK = 0, OH = 0, OW = 0;
    int TH = 0, ni = 0, IHT = 0, IWP = 0, ROWMUL = 1, COLMUL = 1;
    size_t lds = 0;
    size_t wt_bytes = 0, ep_off = 0, ws_off = 0, total = 0;
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static MfmaPlan make_plan(const qe_conv_shape *sh)
{
    MfmaPlan p;
    p.OH = (sh->H + 2 * sh->padding - sh->KH) / sh->stride + 1;
    p.OW = (sh->W + 2 * sh->padding - sh->KW) / sh->stride + 1;
    p.KK = sh->KH * sh->KW;
    if (p.OH <= 0 || p.OW <= 0 || sh->N <= 0 || sh->OC <= 0) return p;
    if ((int64_t)sh->IC * sh->H * sh->W >= (1ll << 31)) return p;
    p.cfg = sh->OC > 64 ? 0 : (sh->OC > 32 ? 1 : 2);
    p.MT = p.cfg == 0 ? 128 : (p.cfg == 1 ? 64 : 32);
    // 4x1 waves: 7 column tiles per wave (112 accumulator registers) keeps the 3x3 variant, which
    // also holds 9 weight fragments, inside 256 VGPRs; 224 pixels = 4 rows of 56 / 8 of 28 / 14x14+.
    const int max_tiles = p.cfg == 0 ? 7 : MF_MAX_NTILES;
    if (p.KK > 64 || p.OW > 32 * max_tiles) return p;
    p.OCP = (sh->OC + p.MT - 1) / p.MT * p.MT;
    p.NCH = (sh->IC + 31) / 32;
    p.NG = 2 * p.NCH;
    p.ROWMUL = (sh->KH == 1) ? sh->stride : 1;   // 1xK strided: only every stride-th row is ever read
    p.COLMUL = (sh->KW == 1) ? sh->stride : 1;
    const int NQ = (sh->W + 3) / 4;
    int TH = std::min(p.OH, (32 * max_tiles) / p.OW);
    for (; TH >= 1; --TH) {
        const int IHT = (p.ROWMUL > 1) ? TH : (TH - 1) * sh->stride + sh->KH;
        const int IWP = (p.COLMUL > 1) ? p.OW : (p.OW - 1) * sh->stride + sh->KW;
        const size_t lds = (size_t)2 * IHT * IWP * 16;
        if (lds <= (size_t)MF_MAX_LDS && 2 * IHT * NQ <= MF_UNITS * MF_THREADS) {
            p.TH = TH; p.IHT = IHT; p.IWP = IWP; p.lds = lds;
            break;
        }
    }
    if (p.TH == 0) return p;
    p.ni = (p.TH * p.OW + 31) / 32;
    p.wt_bytes = (size_t)p.KK * p.NG * p.OCP * 16;
    p.ep_off = align_up(p.wt_bytes, 256);
    p.ws_off = align_up(p.ep_off + (size_t)3 * p.OCP * sizeof(float), 256);
    p.total = align_up(p.ws_off + (size_t)p.OCP * (p.KK + 1) * sizeof(int), 256);
    p.ok = true;
    return p;
}

bool mfma_conv_eligible(const qe_conv_shape *sh, const qe_qparam *x, const qe_qparam *w)
{
    (void)w;
    if (x->n_param != 1) return false;  // per-channel activation scale cannot leave the K sum
    return make_plan(sh).ok;
}

size_t mfma_conv_workspace_bytes(const qe_conv_shape *sh)
{
    const MfmaPlan p = make_plan(sh);
    return p.ok ? p.total : 0;
}

template <int WM, int WN, int NIW>
static void launch_cfg(const MfmaArgs &a, int KK, unsigned blocks, size_t lds, hipStream_t s)
{
    if (KK == 1)
        hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, NIW, 1>), dim3(blocks), dim3(MF_THREADS), lds, s, a);
    else if (KK == 9 && a.KW == 3)
        hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, NIW, 9>), dim3(blocks), dim3(MF_THREADS), lds, s, a);
    else
        hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, NIW, 0>), dim3(blocks), dim3(MF_THREADS), lds, s, a);
}

int launch_conv_mfma(const qe_qparam *x, const qe_qparam *w, const float *bias, const qe_conv_shape *sh,
                     float *out, void *workspace, size_t workspace_bytes, hipStream_t s)
{
    const MfmaPlan p = make_plan(sh);
    if (!p.ok) return QE_ERR_UNSUPPORTED;
    if (workspace == nullptr || workspace_bytes < p.total) return QE_ERR_WORKSPACE;
    if ((reinterpret_cast<uintptr_t>(workspace) & 15) != 0) return QE_ERR_ARG;
    uint8_t *wsp = static_cast<uint8_t *>(workspace);

    PrepArgs pa;
    pa.w = w->data; pa.w_scale = w->scale; pa.w_zero = w->zero; pa.x_scale = x->scale; pa.bias = bias;
    pa.w_bits = w->n_bits; pa.w_sign = w->sign; pa.w_per_tensor = (w->n_param == 1);
    pa.OC = sh->OC; pa.IC = sh->IC; pa.KK = p.KK; pa.OCP = p.OCP; pa.NG = p.NG;
    pa.wt = reinterpret_cast<int8_t *>(wsp);
    pa.ep = reinterpret_cast<float *>(wsp + p.ep_off);
    pa.ws = reinterpret_cast<int *>(wsp + p.ws_off);
    hipLaunchKernelGGL(conv_mfma_prep_kernel, dim3(p.OCP), dim3(256), 0, s, pa);
    QE_LAUNCH_CHECK();

    MfmaArgs a;
    a.x = x->data;
    a.x_bytes = qe_packed_nbytes((int64_t)sh->N * sh->IC * sh->H * sh->W, x->n_bits);
    a.x_zero = x->zero; a.x_bits = x->n_bits; a.x_sign = x->sign;
    a.wt = pa.wt; a.ep = pa.ep; a.ws = pa.ws; a.out = out;
    a.N = sh->N; a.IC = sh->IC; a.H = sh->H; a.W = sh->W; a.OC = sh->OC; a.KH = sh->KH; a.KW = sh->KW;
    a.stride = sh->stride; a.pad = sh->padding; a.OH = p.OH; a.OW = p.OW;
    a.OCP = p.OCP; a.NG = p.NG; a.NCH = p.NCH;
    a.TH = p.TH; a.tiles_h = (p.OH + p.TH - 1) / p.TH;
    a.n_pix_tiles = sh->N * a.tiles_h;
    a.n_oc_tiles = p.OCP / p.MT;
    a.IHT = p.IHT; a.IWP = p.IWP; a.ROWMUL = p.ROWMUL; a.COLMUL = p.COLMUL; a.ni = p.ni;

    const int64_t groups = ((int64_t)a.n_pix_tiles + 7) / 8;
    const int64_t blocks = groups * 8 * a.n_oc_tiles;
    if (blocks > 0x7fffffffLL) return QE_ERR_UNSUPPORTED;
    switch (p.cfg) {
        case 0: launch_cfg<4, 1, 7>(a, p.KK, (unsigned)blocks, p.lds, s); break;
        case 1: launch_cfg<2, 2, 4>(a, p.KK, (unsigned)blocks, p.lds, s); break;
        default: launch_cfg<1, 4, 2>(a, p.KK, (unsigned)blocks, p.lds, s); break;
    }
    QE_LAUNCH_CHECK();
    return QE_OK;
}

}  // namespace qe
