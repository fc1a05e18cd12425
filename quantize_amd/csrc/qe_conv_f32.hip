// qe_conv_f32.hip -- quantconv2d_float_input on the matrix cores (gfx950).
//
// Replaces quantconv2d_float_input_cuda (engine/kernels/functions/quantconv2d_float_input.cu:45-121): fp32 NCHW
// activations x packed b-bit weights, one thread per output, weight unpack + dequantisation per tap, fp32 FMA.  It is
// the form the reference's own module names first (modelzoo/modules/quantconv2d.py:200-203, "only support float input
// and packed weight"), and round 1 served it only with the order-preserving VALU kernel (qe_conv_generic.hip, 13-33
// TFLOP/s: 20-40x the packed-int path).
//
//   out[n,oc,p] = bias[oc] + sum_inb x (q_w - z_w[oc]) s_w[oc] = bias + s_w ( S_xq - z_w S_x )
//   S_xq = sum_inb x q_w   on v_mfma_f32_32x32x16_bf16 with an EXACT split of the activations:
//          x = x1 + x2 + x3, every part a bf16 (8 significant bits each = the 24 of an fp32: x1 = x truncated to
//          bf16, x2 = (x - x1) truncated, x3 = x - x1 - x2, all exact); the integer codes q_w (|q| <= 255) are exact
//          in bf16, so every product x_i q is exact in the fp32 accumulator and only the accumulation rounds -- like
//          the reference's own fp32 chain, in another order.  Three MFMAs per 16-deep k-step instead of one: 6x the
//          matrix time of the int8 path, which leaves this operator HBM-bound (fp32 in + fp32 out) on every
//          ResNet-50 layer but the 3x3 ones.
//   S_x  = sum_inb x           only where some z_w != 0 (asymmetric weights): per-input-pixel channel sums kept by the
//          staging threads in registers (deterministic), summed over the in-bounds taps in the epilogue.  Padded taps
//          are SKIPPED by the reference (quantconv2d_float_input.cu:96): the halo image holds zeros there.
//
// Structure = the halo kernel of the packed path (qe_conv_mfma_kernel.hpp, conv_mfma_kernel): implicit GEMM
// D[oc, pixel], tile = MT output channels x (rows x full width) pixels, LDS halo image pixel-major with zeroed borders
// so a tap is a constant LDS offset (no im2col), any K x K / stride / padding.  What differs: a k-step is 16 channels;
// the image holds per halo pixel 3 splits x 2 k-halves x 8 bf16; a staging thread owns (k-half, halo row, 4 columns):
// 8 coalesced 16-byte loads (one per channel), the splits and the channel-major -> pixel-major turn in registers,
// 12 ds_write_b128.  Weight fragments (bf16, prepared once per call or kept by the caller) go L2 -> VGPR one tap ahead.
#include "qe_common.h"

#include <algorithm>
#include <cstdlib>

namespace qe {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

constexpr int F32_THREADS = 256;
constexpr int F32_TRASH = 64;
constexpr int F32_MAX_LDS = 64 * 1024;

struct F32Args {
    const float *x;
    const uint16_t *wt;        // bf16 [KK][NG][OCP][16]
    const float *ep;           // [3][OCP]: sw, zw, bias
    float *out;
    int N, IC, H, W, OC, KH, KW, stride, pad, OH, OW;
    int OCP, NG;               // padded oc, 16-channel groups
    int TH, tiles_h, n_pix_tiles, n_oc_tiles, GI;
    int IHT, IWP, ROWMUL, COLMUL;
    int chunk;
};

struct F32PrepArgs {
    const uint8_t *w;
    const float *w_scale, *w_zero, *bias;
    int w_bits, w_sign, w_per_tensor;
    int OC, IC, KK, OCP, NG;
    uint16_t *wt;
    float *ep;
};

__device__ __forceinline__ int f32_unpack_code(const uint8_t *__restrict__ p, int64_t ele_idx, int n_bits)
{
    const int64_t bit = ele_idx * n_bits;
    const int64_t byte_idx = bit >> 3;
    const int bit_idx = (int)(bit & 7);
    unsigned v = ((unsigned)p[byte_idx] >> bit_idx);
    if (bit_idx + n_bits > 8) v |= ((unsigned)p[byte_idx + 1] << (8 - bit_idx));
    return (int)(v & ((1u << n_bits) - 1u));
}

// one workgroup per (padded) output channel: q_w as bf16 in fragment order + per-channel constants
__global__ __launch_bounds__(256) void conv_f32_prep_kernel(const F32PrepArgs a)
{
    const int oc = blockIdx.x, tid = threadIdx.x;
    const bool live = oc < a.OC;
    const int off = a.w_sign ? (1 << (a.w_bits - 1)) : 0;       // stored code u = q + off (quantconv2d_float_input.cu:94-101)
    for (int idx = tid; idx < a.KK * a.NG; idx += 256) {
        const int tap = idx / a.NG, g = idx - tap * a.NG;
        uint32_t v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (live) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int ic = g * 16 + j;
                if (ic < a.IC) {
                    const int q = f32_unpack_code(a.w, ((int64_t)oc * a.IC + ic) * a.KK + tap, a.w_bits) - off;
                    const uint32_t b = __float_as_uint((float)q) >> 16;          // |q| <= 255: exact in bf16
                    v[j >> 1] |= b << (16 * (j & 1));
                }
            }
        }
        uint4 *dst = reinterpret_cast<uint4 *>(a.wt + (((int64_t)tap * a.NG + g) * a.OCP + oc) * 16);
        dst[0] = make_uint4(v[0], v[1], v[2], v[3]);
        dst[1] = make_uint4(v[4], v[5], v[6], v[7]);
    }
    if (tid == 0) {
        float sw = 0.0f, zw = 0.0f, b = 0.0f;
        if (live) {
            sw = a.w_per_tensor ? a.w_scale[0] : a.w_scale[oc];
            zw = a.w_per_tensor ? a.w_zero[0] : a.w_zero[oc];
            b = a.bias ? a.bias[oc] : 0.0f;
        }
        a.ep[oc] = sw;
        a.ep[a.OCP + oc] = zw;
        a.ep[2 * a.OCP + oc] = b;
    }
}

// NS = 16-channel groups per stage: 2 where 4 x (units per k-half) <= 256 threads and the LDS allows (every 1x1 layer:
// a 224-pixel tile is 56 row-quads): half the stages, i.e. half the exposed memory round trips of the K loop.
template <int WM, int WN, int NIW, int NS>
__global__ __launch_bounds__(F32_THREADS, 2) void conv_f32_mfma_kernel(const F32Args a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint4 *Xs = reinterpret_cast<uint4 *>(smem);
    constexpr int MT = 32 * WM;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int col = lane & 31, h = lane >> 5;

    // ---- tile decode (XCD-aware block map, as block_to_tile of the packed kernels) ----------------------------
    int pt, ot;
    {
        const int bid = blockIdx.x, idx = bid >> 3;
        const int j = idx / a.n_oc_tiles;
        ot = idx - j * a.n_oc_tiles;
        const int c = j / a.chunk;
        pt = (c * 8 + (bid & 7)) * a.chunk + (j - c * a.chunk);
    }
    if (pt >= a.n_pix_tiles) return;
    const int ng = pt / a.tiles_h;
    const int n0 = ng * a.GI;
    const int oh0 = (pt - ng * a.tiles_h) * a.TH;
    const int th = min(a.TH, a.OH - oh0);
    const int OHWt = th * a.OW;
    const int NT = min(a.GI, a.N - n0) * OHWt;
    const int ih0 = oh0 * a.stride - a.pad;
    const int ISZ = a.IHT * a.IWP, GSZ = a.GI * ISZ;
    const int KK = a.KH * a.KW;
    const int trash = 6 * NS * GSZ + lane;
    float *sxp = reinterpret_cast<float *>(Xs + 6 * NS * GSZ + F32_TRASH);      // [2 NS][GSZ] per-input-pixel channel sums of each staging slice

    for (int i = tid; i < 6 * NS * GSZ; i += F32_THREADS) Xs[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < 2 * NS * GSZ; i += F32_THREADS) sxp[i] = 0.0f;

    // ---- per-lane pixel bases of the wave's column tiles (uint4 index; split s adds 2 s GSZ) -------------------
    const int RS = a.stride / a.ROWMUL, CS = a.stride / a.COLMUL;
    int pixidx[NIW];
#pragma unroll
    for (int t = 0; t < NIW; ++t) {
        const int q = (wn + t * WN) * 32 + col;
        const int gi = (a.GI > 1) ? q / OHWt : 0;
        const int rq = q - gi * OHWt;
        const int r = rq / a.OW, c = rq - r * a.OW;
        pixidx[t] = h * GSZ + ((q < NT) ? gi * ISZ + (r * RS) * a.IWP + c * CS : 0);
    }

    // ---- staging unit of this thread: (k-half hh, image gi, halo row l, column quad iq) -------------------------
    const int NQ = (a.W + 3) >> 2;
    const int U = a.GI * a.IHT * NQ;                 // units per k-half (host: 2 U <= 256)
    const int HW = a.H * a.W;
    const int sub = (NS == 2 && tid >= 2 * U) ? 1 : 0;          // which group of the stage this thread stages
    const int tsub = tid - sub * 2 * U;
    const int hh = tsub >= U ? 1 : 0;
    int u_off = 0, u_es = 0, u_lds[4];
    bool u_live;
    {
        const int lt = tsub - hh * U;
        const int gi = lt / (a.IHT * NQ);
        const int rr = lt - gi * (a.IHT * NQ);
        const int l = rr / NQ, iq = rr - l * NQ;
        const int ih = ih0 + l * a.ROWMUL;
        u_live = lt < U && tsub < 2 * U;
        const bool ok = u_live && gi < a.GI && n0 + gi < a.N && ih >= 0 && ih < a.H;
        int iw0 = 4 * iq;
        if (iw0 + 4 > a.W) { u_es = iw0 + 4 - a.W; iw0 = a.W - 4; }       // never read past the row (W >= 4): shifted back
        u_off = ok ? gi * a.IC * HW + ih * a.W + iw0 : 0;
        if (!ok) u_es = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int iw = 4 * iq + j;
            const int cl = iw + a.pad;
            const int clc = cl / a.COLMUL;
            const bool pok = ok && iw < a.W && (clc * a.COLMUL == cl) && clc < a.IWP;
            u_lds[j] = pok ? (gi * a.IHT + l) * a.IWP + clc : -1;
        }
    }
    const float *xi = a.x + (int64_t)n0 * a.IC * HW;

    // ---- weight fragments: lane (row col of strip wm, k-half h) reads 8 bf16 of Wt[tap][g][oc][16] -------------
    const uint16_t *a_base = a.wt + ((int64_t)(ot * MT + wm * 32 + col) * 16 + 8 * h);
    const int64_t grp_stride = (int64_t)a.OCP * 16;            // elements
    const int64_t tap_stride = (int64_t)a.NG * grp_stride;

    v16f acc[NIW];
#pragma unroll
    for (int t = 0; t < NIW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    int zw_local = 0;
    if (tid < MT) zw_local = (a.ep[a.OCP + ot * MT + tid] != 0.0f) ? 1 : 0;
    const bool need_sx = __syncthreads_or(zw_local) != 0;      // also orders the LDS zero fill

    float4 d[8];
    float sx_priv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    auto issue_x = [&](int st) __attribute__((always_inline)) {
        const int g = st * NS + sub;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ic = g * 16 + hh * 8 + i;
            const int icc = ic < a.IC ? ic : a.IC - 1;
            __builtin_memcpy(&d[i], xi + ((int64_t)icc * HW + u_off), 16);     // 4-byte aligned global_load_dwordx4
        }
    };
    auto stage_x = [&](int st) __attribute__((always_inline)) {
        const int g = st * NS + sub;
        // pixel j of the quad sits at element j + u_es of the (shifted-back) load: rotate once per channel
        float xr[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool pad_ch = g * 16 + hh * 8 + i >= a.IC;                  // channel padding of the last group
            const float e0 = pad_ch ? 0.0f : d[i].x, e1 = pad_ch ? 0.0f : d[i].y, e2 = pad_ch ? 0.0f : d[i].z, e3 = pad_ch ? 0.0f : d[i].w;
            xr[i][0] = u_es == 0 ? e0 : (u_es == 1 ? e1 : (u_es == 2 ? e2 : e3));
            xr[i][1] = u_es == 0 ? e1 : (u_es == 1 ? e2 : e3);
            xr[i][2] = u_es == 0 ? e2 : e3;
            xr[i][3] = e3;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x[8];
            float s = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                x[i] = xr[i][j];
                s += x[i];
            }
            if (need_sx) sx_priv[j] += s;
            const bool live = u_lds[j] >= 0;
            const int idx = live ? u_lds[j] + (sub * 6 + hh) * GSZ : trash;
            // one split at a time: the truncated part goes to LDS (8 channels x bf16 = one 16-byte vector: dword m holds
            // channels 2m (low half) and 2m + 1 (high half)), the exact remainder stays in x
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) {
                uint32_t u[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    u[i] = __float_as_uint(x[i]) & 0xffff0000u;               // sp == 2: the remainder has <= 8 significant bits
                    x[i] -= __uint_as_float(u[i]);
                }
                Xs[live ? idx + 2 * sp * GSZ : trash] =
                    make_uint4(__builtin_amdgcn_perm(u[1], u[0], 0x07060302u), __builtin_amdgcn_perm(u[3], u[2], 0x07060302u),
                               __builtin_amdgcn_perm(u[5], u[4], 0x07060302u), __builtin_amdgcn_perm(u[7], u[6], 0x07060302u));
            }
        }
    };

    const int n_stages = a.NG / NS;             // NG is padded to a multiple of NS by the host (zero weights)
    issue_x(0);
    for (int st = 0; st < n_stages; ++st) {
        const uint16_t *a_g = a_base + (int64_t)(st * NS) * grp_stride;
        v4i af = *reinterpret_cast<const v4i *>(a_g);                        // group 0 / tap 0, requested before the staging work
        stage_x(st);
        __syncthreads();
        if (st + 1 < n_stages) issue_x(st + 1);
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            for (int tap = 0; tap < KK; ++tap) {
                // next fragment: next tap of this group, or tap 0 of the stage's second group
                const bool last_tap = tap + 1 == KK;
                const uint16_t *nx = (last_tap && k + 1 < NS) ? a_g + (int64_t)(k + 1) * grp_stride
                                                                : a_g + (int64_t)k * grp_stride + (int64_t)(last_tap ? tap : tap + 1) * tap_stride;
                const v4i af_next = *reinterpret_cast<const v4i *>(nx);
                const int kh = tap / a.KW;
                const int off = k * 6 * GSZ + kh * a.IWP + (tap - kh * a.KW);
                const v8bf wf = __builtin_bit_cast(v8bf, af);
                // the three split fragments of column tile t + 1 are requested before the MFMAs of tile t (left alone,
                // hipcc keeps two fragment registers and waits for every read right in front of its MFMA)
                v4i bc[3], bn[3];
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) bc[sp] = *reinterpret_cast<const v4i *>(&Xs[pixidx[0] + off + 2 * sp * GSZ]);
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);       // the 3 reads of tile 0
#pragma unroll
                for (int t = 0; t < NIW; ++t) {
                    if (t + 1 < NIW) {
#pragma unroll
                        for (int sp = 0; sp < 3; ++sp) bn[sp] = *reinterpret_cast<const v4i *>(&Xs[pixidx[t + 1] + off + 2 * sp * GSZ]);
                    }
#pragma unroll
                    for (int sp = 2; sp >= 0; --sp)                              // smallest parts first
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, __builtin_bit_cast(v8bf, bc[sp]), acc[t], 0, 0, 0);
#pragma unroll
                    for (int sp = 0; sp < 3; ++sp) bc[sp] = bn[sp];
                    if (t + 1 < NIW) __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);   // 3 DS reads (tile t + 1) ...
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                    // ... ahead of the 3 MFMAs of tile t
                }
                af = af_next;
            }
        }
        __syncthreads();
    }

    // ---- epilogue: D rows = output channel (register), cols = pixel (lane) ------------------------------------
    float sxs[NIW];
#pragma unroll
    for (int t = 0; t < NIW; ++t) sxs[t] = 0.0f;
    if (need_sx) {
        // every staging thread owns one (group-of-the-stage, k-half) slice of its pixels: private slots, summed in a
        // fixed order (no atomics: the result does not depend on the order the threads arrive in)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (u_lds[j] >= 0) sxp[(sub * 2 + hh) * GSZ + u_lds[j]] = sx_priv[j];
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NIW; ++t) {
            const int pbase = pixidx[t] - h * GSZ;
            for (int tap = 0; tap < KK; ++tap) {
                const int kh = tap / a.KW;
                const int o = pbase + kh * a.IWP + (tap - kh * a.KW);           // zero outside the image
                float v = sxp[o] + sxp[GSZ + o];
                if constexpr (NS == 2) v += sxp[2 * GSZ + o] + sxp[3 * GSZ + o];
                sxs[t] += v;
            }
        }
    }
    const int oc_base = ot * MT + wm * 32 + 4 * h;
    float sw[16], zw[16], bi[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int oc = oc_base + (r & 3) + 8 * (r >> 2);
        sw[r] = a.ep[oc];
        zw[r] = a.ep[a.OCP + oc];
        bi[r] = a.ep[2 * a.OCP + oc];
    }
    const int OHW = a.OH * a.OW;
    const bool full_oc = (ot + 1) * MT <= a.OC;
    float *out_w = a.out + ((int64_t)n0 * a.OC + ot * MT + wm * 32) * OHW + (int64_t)oh0 * a.OW;
    // 16-byte stores need one image per tile, 16-byte aligned rows and the halo image at least as large as the 4 patches
    const bool use_patch = a.GI == 1 && (OHW & 3) == 0 && ((oh0 * a.OW) & 3) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0 &&
                           6 * NS * GSZ * 16 >= 4 * 32 * 36 * 4 && !(need_sx && false);
    float *patch = reinterpret_cast<float *>(smem) + wave * (32 * 36);
#pragma unroll
    for (int t = 0; t < NIW; ++t) {
        const int q = (wn + t * WN) * 32 + col;
        const int gi = (a.GI > 1) ? q / OHWt : 0;
        const int rq = q - gi * OHWt;
        const bool valid = q < NT;
        const uint32_t voff = valid ? (uint32_t)(gi * a.OC + 4 * h) * (uint32_t)OHW + (uint32_t)rq : 0u;
        const int q0 = (wn + t * WN) * 32;
        const bool whole = full_oc && q0 + 32 <= NT;            // wave-uniform: plain stores
        if (whole && use_patch) {
            // whole 32 x 32 tile of one image: through the wave's LDS patch (the halo image is dead) and out as 16-byte
            // pieces, 8 rows x 128 contiguous bytes per store instruction instead of 2 x 128 (the write-bound 1x1 layers
            // stored at 3.5 TB/s against 5.1 TB/s on the packed path, which has stored this way since round 1)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = (r & 3) + 8 * (r >> 2);
                float v = acc[t][r];
                if (need_sx) v = fmaf(-zw[r], sxs[t], v);
                patch[(dr + 4 * h) * 36 + col] = fmaf(sw[r], v, bi[r]);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): same wave wrote and reads
            float *tile = out_w + q0 + 4 * (lane & 7);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int rt = 8 * k + (lane >> 3);
                const float4 o4 = *reinterpret_cast<const float4 *>(patch + rt * 36 + 4 * (lane & 7));
                *reinterpret_cast<float4 *>(tile + (int64_t)rt * OHW) = o4;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);   // reads done before the next tile overwrites the patch
            continue;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            float v = acc[t][r];
            if (need_sx) v = fmaf(-zw[r], sxs[t], v);
            const float res = fmaf(sw[r], v, bi[r]);
            if (whole || (valid && oc_base + dr < a.OC)) (out_w + (int64_t)dr * OHW)[voff] = res;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Small-IC variant (IC <= 4, KH, KW <= 8): the stem (3 -> 64, 7x7 / 2).  Padding 3 channels to a 16-channel k-step
// would waste 5x the MFMA work and the halo image of a 224-wide row band would not fit the LDS, so K is laid out as in the
// packed stem kernel (conv_mfma_smallic_kernel): per kernel row kh, K = [kw 0..7][ic 0..3] = two 16-deep k-steps.  LDS
// holds, per split, one 8-byte vector [x_ic0 x_ic1 x_ic2 0] (bf16) per halo pixel; the B fragment of lane (pixel, half h)
// for k-step j is the 16 bytes of the 2 consecutive pixels at column ow*stride + 4j + 2h of row oh*stride + kh.
//   Wt layout here: [KH * 2][1][OCP][16]: element 8h + 4 kwl + ic of "tap" 2 kh + j = q_w[oc][ic][kh][4j + 2h + kwl].
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void conv_f32_prep_stem_kernel(const F32PrepArgs a, int KH, int KW)
{
    const int oc = blockIdx.x, tid = threadIdx.x;
    const bool live = oc < a.OC;
    const int off = a.w_sign ? (1 << (a.w_bits - 1)) : 0;
    for (int idx = tid; idx < KH * 2; idx += 64) {
        const int kh = idx >> 1, j = idx & 1;
        uint32_t v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (live) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int hh = e >> 3, kwl = (e >> 2) & 1, ic = e & 3;
                const int kw = 4 * j + 2 * hh + kwl;
                if (kw < KW && ic < a.IC) {
                    const int q = f32_unpack_code(a.w, (((int64_t)oc * a.IC + ic) * KH + kh) * KW + kw, a.w_bits) - off;
                    v[e >> 1] |= (__float_as_uint((float)q) >> 16) << (16 * (e & 1));
                }
            }
        }
        uint4 *dst = reinterpret_cast<uint4 *>(a.wt + ((int64_t)idx * a.OCP + oc) * 16);
        dst[0] = make_uint4(v[0], v[1], v[2], v[3]);
        dst[1] = make_uint4(v[4], v[5], v[6], v[7]);
    }
    if (tid == 0) {
        float sw = 0.0f, zw = 0.0f, b = 0.0f;
        if (live) {
            sw = a.w_per_tensor ? a.w_scale[0] : a.w_scale[oc];
            zw = a.w_per_tensor ? a.w_zero[0] : a.w_zero[oc];
            b = a.bias ? a.bias[oc] : 0.0f;
        }
        a.ep[oc] = sw;
        a.ep[a.OCP + oc] = zw;
        a.ep[2 * a.OCP + oc] = b;
    }
}

template <int WM, int WN, int NIW>
__global__ __launch_bounds__(F32_THREADS, 2) void conv_f32_stem_kernel(const F32Args a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint2 *Xs = reinterpret_cast<uint2 *>(smem);                 // [3 splits][GSZ] (+ trash)
    constexpr int MT = 32 * WM;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int col = lane & 31, h = lane >> 5;

    int pt, ot;
    {
        const int bid = blockIdx.x, idx = bid >> 3;
        const int j = idx / a.n_oc_tiles;
        ot = idx - j * a.n_oc_tiles;
        const int c = j / a.chunk;
        pt = (c * 8 + (bid & 7)) * a.chunk + (j - c * a.chunk);
    }
    if (pt >= a.n_pix_tiles) return;
    const int n0 = pt / a.tiles_h;
    const int oh0 = (pt - n0 * a.tiles_h) * a.TH;
    const int th = min(a.TH, a.OH - oh0);
    const int NT = th * a.OW;
    const int ih0 = oh0 * a.stride - a.pad;
    const int GSZ = a.IHT * a.IWP;
    const int trash = 3 * GSZ + lane;
    float *sxp = reinterpret_cast<float *>(Xs + 3 * GSZ + F32_TRASH);           // [GSZ] channel sums per halo pixel

    for (int i = tid; i < 3 * GSZ; i += F32_THREADS) Xs[i] = make_uint2(0, 0);
    for (int i = tid; i < GSZ; i += F32_THREADS) sxp[i] = 0.0f;

    int zw_local = 0;
    if (tid < MT) zw_local = (a.ep[a.OCP + ot * MT + tid] != 0.0f) ? 1 : 0;
    const bool need_sx = __syncthreads_or(zw_local) != 0;      // also orders the LDS zero fill

    // weight fragments of all kernel rows (L2 hits), requested before the activation loads
    v4i afr[8][2];
    {
        const uint16_t *a_base = a.wt + ((int64_t)(ot * MT + wm * 32 + col) * 16 + 8 * h);
#pragma unroll
        for (int kh = 0; kh < 8; ++kh) {
            const int khc = kh < a.KH ? kh : a.KH - 1;
#pragma unroll
            for (int j = 0; j < 2; ++j) afr[kh][j] = *reinterpret_cast<const v4i *>(a_base + (int64_t)(khc * 2 + j) * a.OCP * 16);
        }
    }

    // ---- stage the halo tile: thread <-> (row l, column quad iq): IC 16-byte loads, 4 pixels x 3 splits x 8 bytes ----
    const int NQ = (a.W + 3) >> 2;
    const int HW = a.H * a.W;
    const float *xi = a.x + (int64_t)n0 * a.IC * HW;
    const int n_units = a.IHT * NQ;
    for (int u0 = 0; u0 < n_units; u0 += 2 * F32_THREADS) {
        float4 d[2][4];
        int ul[2], uq[2], ues[2];
        bool uok[2], ulive[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int uid = u0 + tid + it * F32_THREADS;
            ulive[it] = uid < n_units;
            const int uc = ulive[it] ? uid : 0;
            const int l = uc / NQ, iq = uc - l * NQ;
            const int ih = ih0 + l;
            const bool ok = ih >= 0 && ih < a.H;
            int iw0 = 4 * iq, es = 0;
            if (iw0 + 4 > a.W) { es = iw0 + 4 - a.W; iw0 = a.W - 4; }          // never read past the row: shifted back
            const uint32_t off = ok ? (uint32_t)(ih * a.W + iw0) : 0u;
            ul[it] = l; uq[it] = iq; ues[it] = es; uok[it] = ok;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int icc = i < a.IC ? i : a.IC - 1;       // uniform
                __builtin_memcpy(&d[it][i], xi + (int64_t)icc * HW + off, 16);
            }
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            float xr[4][4];                                    // [channel][pixel of the quad]
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool pad_ch = i >= a.IC;
                const float e0 = pad_ch ? 0.0f : d[it][i].x, e1 = pad_ch ? 0.0f : d[it][i].y, e2 = pad_ch ? 0.0f : d[it][i].z, e3 = pad_ch ? 0.0f : d[it][i].w;
                const int es = ues[it];
                xr[i][0] = es == 0 ? e0 : (es == 1 ? e1 : (es == 2 ? e2 : e3));
                xr[i][1] = es == 0 ? e1 : (es == 1 ? e2 : e3);
                xr[i][2] = es == 0 ? e2 : e3;
                xr[i][3] = e3;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int iw = 4 * uq[it] + j;
                const int cl = iw + a.pad;
                const bool pok = ulive[it] && uok[it] && iw < a.W && cl < a.IWP;
                const int idx = pok ? ul[it] * a.IWP + cl : -1;
                float x[4] = {xr[0][j], xr[1][j], xr[2][j], xr[3][j]};
                if (need_sx && pok) sxp[idx] = ((x[0] + x[1]) + x[2]) + x[3];
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) {
                    uint32_t u[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        u[i] = __float_as_uint(x[i]) & 0xffff0000u;           // sp == 2: the remainder has <= 8 significant bits
                        x[i] -= __uint_as_float(u[i]);
                    }
                    Xs[pok ? idx + sp * GSZ : trash] =
                        make_uint2(__builtin_amdgcn_perm(u[1], u[0], 0x07060302u), __builtin_amdgcn_perm(u[3], u[2], 0x07060302u));
                }
            }
        }
    }
    __syncthreads();

    int pixidx[NIW];
#pragma unroll
    for (int t = 0; t < NIW; ++t) {
        const int q = (wn + t * WN) * 32 + col;
        const int r = q / a.OW, c = q - r * a.OW;
        pixidx[t] = (q < NT) ? (r * a.stride) * a.IWP + c * a.stride : 0;
    }
    v16f acc[NIW];
#pragma unroll
    for (int t = 0; t < NIW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
#pragma unroll
    for (int kh = 0; kh < 8; ++kh) {
        if (kh < a.KH) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const v8bf wf = __builtin_bit_cast(v8bf, afr[kh][j]);
#pragma unroll
                for (int t = 0; t < NIW; ++t) {
                    const uint2 *bp = &Xs[pixidx[t] + kh * a.IWP + 4 * j + 2 * h];
#pragma unroll
                    for (int sp = 2; sp >= 0; --sp) {                            // smallest parts first
                        const uint2 b0 = bp[sp * GSZ], b1 = bp[sp * GSZ + 1];
                        const v4i b = {(int)b0.x, (int)b0.y, (int)b1.x, (int)b1.y};
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, __builtin_bit_cast(v8bf, b), acc[t], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- epilogue (as conv_f32_mfma_kernel) ----
    float sxs[NIW];
#pragma unroll
    for (int t = 0; t < NIW; ++t) {
        sxs[t] = 0.0f;
        if (need_sx) {
            for (int kh = 0; kh < a.KH; ++kh)
                for (int kw = 0; kw < a.KW; ++kw) sxs[t] += sxp[pixidx[t] + kh * a.IWP + kw];     // zero outside the image
        }
    }
    const int oc_base = ot * MT + wm * 32 + 4 * h;
    float sw[16], zw[16], bi[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int oc = oc_base + (r & 3) + 8 * (r >> 2);
        sw[r] = a.ep[oc];
        zw[r] = a.ep[a.OCP + oc];
        bi[r] = a.ep[2 * a.OCP + oc];
    }
    const int OHW = a.OH * a.OW;
    const bool full_oc = (ot + 1) * MT <= a.OC;
    float *out_w = a.out + ((int64_t)n0 * a.OC + ot * MT + wm * 32) * OHW + (int64_t)oh0 * a.OW;
    // whole tiles go through the wave's LDS patch and out as 16-byte pieces (as in conv_f32_mfma_kernel); the barrier makes
    // sure every wave has read its last fragments from the halo image the patches overwrite
    const bool use_patch = (OHW & 3) == 0 && ((oh0 * a.OW) & 3) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0 &&
                           3 * GSZ * 8 >= 4 * 32 * 36 * 4;
    float *patch = reinterpret_cast<float *>(smem) + wave * (32 * 36);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NIW; ++t) {
        const int q = (wn + t * WN) * 32 + col;
        const bool valid = q < NT;
        const uint32_t voff = valid ? (uint32_t)(4 * h) * (uint32_t)OHW + (uint32_t)q : 0u;
        const bool whole = full_oc && (wn + t * WN) * 32 + 32 <= NT;            // wave-uniform: plain stores
        if (whole && use_patch) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = (r & 3) + 8 * (r >> 2);
                float v = acc[t][r];
                if (need_sx) v = fmaf(-zw[r], sxs[t], v);
                patch[(dr + 4 * h) * 36 + col] = fmaf(sw[r], v, bi[r]);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): same wave wrote and reads
            float *tile = out_w + (wn + t * WN) * 32 + 4 * (lane & 7);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int rt = 8 * k + (lane >> 3);
                const float4 o4 = *reinterpret_cast<const float4 *>(patch + rt * 36 + 4 * (lane & 7));
                *reinterpret_cast<float4 *>(tile + (int64_t)rt * OHW) = o4;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            continue;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dr = (r & 3) + 8 * (r >> 2);
            float v = acc[t][r];
            if (need_sx) v = fmaf(-zw[r], sxs[t], v);
            const float res = fmaf(sw[r], v, bi[r]);
            if (whole || (valid && oc_base + dr < a.OC)) (out_w + (int64_t)dr * OHW)[voff] = res;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct F32Plan {
    bool ok = false;
    int cfg = 0;               // 0: 4x1 waves x 7 column tiles (MT 128), 1: 2x2 x 4 (MT 64)
    int MT = 0, OCP = 0, NG = 0, KK = 0, OH = 0, OW = 0;
    int TH = 0, GI = 1, IHT = 0, IWP = 0, ROWMUL = 1, COLMUL = 1, NS = 1;
    bool stem = false;         // IC <= 4: K = kh x [kw 0..7][ic 0..3] (conv_f32_stem_kernel)
    int niw = 0;
    size_t lds = 0, wt_bytes = 0, ep_off = 0, total = 0;
};

static size_t f32_align(size_t v, size_t a) { return (v + a - 1) / a * a; }

static F32Plan f32_plan(const qe_conv_shape *sh)
{
    F32Plan p;
    if (env_get("QE_F32_MFMA") && atoi(env_get("QE_F32_MFMA")) == 0) return p;
    p.OH = (sh->H + 2 * sh->padding - sh->KH) / sh->stride + 1;
    p.OW = (sh->W + 2 * sh->padding - sh->KW) / sh->stride + 1;
    p.KK = sh->KH * sh->KW;
    if (p.OH <= 0 || p.OW <= 0 || sh->N <= 0 || sh->OC <= 0) return p;
    if (sh->W < 4 || p.KK > 64) return p;
    if (sh->IC <= 4 && sh->KH <= 8 && sh->KW <= 8 && !(env_get("QE_F32_STEM") && atoi(env_get("QE_F32_STEM")) == 0)) {
        // the stem: K = kh x [kw][ic]; tile = whole output rows, as many as 448 / 224 pixel slots and 64 KB of LDS hold
        if ((int64_t)sh->IC * sh->H * sh->W >= (1ll << 29) || (int64_t)sh->OC * p.OH * p.OW >= (1ll << 29)) return p;
        p.stem = true;
        p.cfg = sh->OC > 64 ? 0 : 1;
        p.MT = p.cfg == 0 ? 128 : 64;
        p.OCP = (sh->OC + p.MT - 1) / p.MT * p.MT;
        p.NG = 1;
        const int max_tiles = p.cfg == 0 ? 7 : 14;
        if (p.OW > 32 * max_tiles) { p.stem = false; return p; }
        for (int TH = std::min(p.OH, 32 * max_tiles / p.OW); TH >= 1; --TH) {
            const int IHT = (TH - 1) * sh->stride + sh->KH, IWP = (p.OW - 1) * sh->stride + 8;
            const size_t gsz = (size_t)IHT * IWP;
            const size_t lds = (3 * gsz + F32_TRASH) * 8 + gsz * 4;
            if (lds <= (size_t)F32_MAX_LDS) { p.TH = TH; p.IHT = IHT; p.IWP = IWP; p.lds = lds; break; }
        }
        if (p.TH == 0) { p.stem = false; return p; }
        const int ni = (p.TH * p.OW + 31) / 32;
        p.niw = p.cfg == 0 ? 7 : (ni <= 8 ? 4 : 7);          // 2 x 2 waves: 8 or 14 column slots
        p.wt_bytes = (size_t)sh->KH * 2 * p.OCP * 16 * sizeof(uint16_t);
        p.ep_off = f32_align(p.wt_bytes, 256);
        p.total = f32_align(p.ep_off + (size_t)3 * p.OCP * sizeof(float), 256);
        p.ok = true;
        return p;
    }
    if (sh->IC < 8) return p;                                               // 5..7 channels stay on the VALU kernel
    if ((int64_t)sh->IC * sh->H * sh->W * 8 >= (1ll << 31)) return p;       // 32-bit element offsets inside a tile's (<= 8) images
    if ((int64_t)sh->OC * p.OH * p.OW >= (1ll << 29)) return p;
    p.cfg = sh->OC > 64 ? 0 : 1;
    // 3x3 on 7x7 maps: 64-channel workgroups (0.310 -> 0.247 ms on 512->512; every other layer is faster with 128)
    if (sh->OC > 64 && p.KK == 9 && sh->stride == 1 && p.OH * p.OW <= 64) p.cfg = 1;
    if (env_get("QE_F32_CFG")) p.cfg = atoi(env_get("QE_F32_CFG")) ? 1 : 0;        // tuning
    p.MT = p.cfg == 0 ? 128 : 64;
    const int max_tiles = p.cfg == 0 ? 7 : 8;
    if (p.OW > 32 * max_tiles) return p;
    p.OCP = (sh->OC + p.MT - 1) / p.MT * p.MT;
    p.NG = (sh->IC + 15) / 16;
    p.ROWMUL = (sh->KH == 1) ? sh->stride : 1;
    p.COLMUL = (sh->KW == 1) ? sh->stride : 1;
    const int NQ = (sh->W + 3) / 4;
    const int max_px = 32 * max_tiles;
    if (p.OH * p.OW <= max_px / 2) p.GI = std::max(1, std::min((int)sh->N, max_px / (p.OH * p.OW)));
    int TH = (p.GI > 1) ? p.OH : std::min(p.OH, max_px / p.OW);
    for (;;) {
        const int IHT = (p.ROWMUL > 1) ? TH : (TH - 1) * sh->stride + sh->KH;
        const int IWP = (p.COLMUL > 1) ? p.OW : (p.OW - 1) * sh->stride + sh->KW;
        const int units = p.GI * IHT * NQ;
        const size_t gsz = (size_t)p.GI * IHT * IWP;
        const size_t lds = (6 * gsz + F32_TRASH) * 16 + 2 * gsz * 4;
        if (lds <= (size_t)F32_MAX_LDS && 2 * units <= F32_THREADS) { p.TH = TH; p.IHT = IHT; p.IWP = IWP; p.lds = lds; break; }
        if (p.GI > 1) { --p.GI; continue; }
        if (--TH < 1) return p;
    }
    if (p.GI == 1 && p.TH >= 1) {                       // balanced row tiles
        const int nt = (p.OH + p.TH - 1) / p.TH;
        const int th2 = (p.OH + nt - 1) / nt;
        if (th2 < p.TH) {
            p.TH = th2;
            p.IHT = (p.ROWMUL > 1) ? th2 : (th2 - 1) * sh->stride + sh->KH;
            const size_t gsz = (size_t)p.IHT * p.IWP;
            p.lds = (6 * gsz + F32_TRASH) * 16 + 2 * gsz * 4;
        }
    }
    {   // two groups per stage where threads and LDS allow and the K loop is long enough to matter
        const int units = p.GI * p.IHT * NQ;
        const size_t gsz = (size_t)p.GI * p.IHT * p.IWP;
        const size_t lds2 = (12 * gsz + F32_TRASH) * 16 + 4 * gsz * 4;
        const bool ns2_env = !(env_get("QE_F32_NS") && atoi(env_get("QE_F32_NS")) == 1);
        if (ns2_env && 4 * units <= F32_THREADS && lds2 <= (size_t)F32_MAX_LDS && p.NG >= 4) { p.NS = 2; p.lds = lds2; p.NG = (p.NG + 1) / 2 * 2; }
    }
    p.wt_bytes = (size_t)p.KK * p.NG * p.OCP * 16 * sizeof(uint16_t);
    if ((int64_t)p.wt_bytes >= (1ll << 31)) return p;
    p.ep_off = f32_align(p.wt_bytes, 256);
    p.total = f32_align(p.ep_off + (size_t)3 * p.OCP * sizeof(float), 256);
    p.ok = true;
    return p;
}

bool f32_conv_eligible(const qe_conv_shape *sh, const qe_qparam *w)
{
    (void)w;
    return f32_plan(sh).ok;
}

size_t f32_conv_prepared_bytes(const qe_conv_shape *sh)
{
    const F32Plan p = f32_plan(sh);
    return p.ok ? p.total : 0;
}

// mode 0: prepare into `prepared` and run; 1: prepare only; 2: run on tables prepared earlier
int launch_conv_f32(const float *x, const qe_qparam *w, const float *bias, const qe_conv_shape *sh, float *out,
                    void *prepared, size_t prepared_bytes, hipStream_t s, int mode)
{
    const F32Plan p = f32_plan(sh);
    if (!p.ok) return QE_ERR_UNSUPPORTED;
    if (prepared == nullptr || prepared_bytes < p.total) return QE_ERR_WORKSPACE;
    if ((reinterpret_cast<uintptr_t>(prepared) & 15) != 0) return QE_ERR_ARG;
    uint8_t *wsp = static_cast<uint8_t *>(prepared);
    if (mode != 2) {
        F32PrepArgs pa;
        pa.w = w->data; pa.w_scale = w->scale; pa.w_zero = w->zero; pa.bias = bias;
        pa.w_bits = w->n_bits; pa.w_sign = w->sign; pa.w_per_tensor = (w->n_param == 1);
        pa.OC = sh->OC; pa.IC = sh->IC; pa.KK = p.KK; pa.OCP = p.OCP; pa.NG = p.NG;
        pa.wt = reinterpret_cast<uint16_t *>(wsp);
        pa.ep = reinterpret_cast<float *>(wsp + p.ep_off);
        if (p.stem) hipLaunchKernelGGL(conv_f32_prep_stem_kernel, dim3(p.OCP), dim3(64), 0, s, pa, (int)sh->KH, (int)sh->KW);
        else hipLaunchKernelGGL(conv_f32_prep_kernel, dim3(p.OCP), dim3(256), 0, s, pa);
        QE_LAUNCH_CHECK();
        if (mode == 1) return QE_OK;
    }
    F32Args a;
    a.x = x; a.wt = reinterpret_cast<const uint16_t *>(wsp); a.ep = reinterpret_cast<const float *>(wsp + p.ep_off); a.out = out;
    a.N = sh->N; a.IC = sh->IC; a.H = sh->H; a.W = sh->W; a.OC = sh->OC; a.KH = sh->KH; a.KW = sh->KW;
    a.stride = sh->stride; a.pad = sh->padding; a.OH = p.OH; a.OW = p.OW;
    a.OCP = p.OCP; a.NG = p.NG; a.TH = p.TH; a.tiles_h = (p.OH + p.TH - 1) / p.TH; a.GI = p.GI;
    a.IHT = p.IHT; a.IWP = p.IWP; a.ROWMUL = p.ROWMUL; a.COLMUL = p.COLMUL;
    a.n_pix_tiles = ((sh->N + p.GI - 1) / p.GI) * a.tiles_h;
    a.n_oc_tiles = p.OCP / p.MT;
    const int64_t per_xcd = ((int64_t)a.n_pix_tiles + 7) / 8;
    a.chunk = (int)std::max<int64_t>(1, per_xcd);
    const int64_t runs = ((int64_t)a.n_pix_tiles + a.chunk - 1) / a.chunk;
    const int64_t blocks = (runs + 7) / 8 * a.chunk * 8 * a.n_oc_tiles;
    if (blocks > 0x7fffffffLL) return QE_ERR_UNSUPPORTED;
    if (p.stem) {
        if (p.cfg == 0) hipLaunchKernelGGL((conv_f32_stem_kernel<4, 1, 7>), dim3((unsigned)blocks), dim3(F32_THREADS), p.lds, s, a);
        else if (p.niw == 4) hipLaunchKernelGGL((conv_f32_stem_kernel<2, 2, 4>), dim3((unsigned)blocks), dim3(F32_THREADS), p.lds, s, a);
        else hipLaunchKernelGGL((conv_f32_stem_kernel<2, 2, 7>), dim3((unsigned)blocks), dim3(F32_THREADS), p.lds, s, a);
        QE_LAUNCH_CHECK();
        return QE_OK;
    }
    const int ni = (p.GI * p.TH * p.OW + 31) / 32;           // column tiles the tile really has
#define QE_F32_LAUNCH(WM, WN, NIW)                                                                                              \
    do {                                                                                                                        \
        if (p.NS == 2) hipLaunchKernelGGL((conv_f32_mfma_kernel<WM, WN, NIW, 2>), dim3((unsigned)blocks), dim3(F32_THREADS), p.lds, s, a); \
        else hipLaunchKernelGGL((conv_f32_mfma_kernel<WM, WN, NIW, 1>), dim3((unsigned)blocks), dim3(F32_THREADS), p.lds, s, a);           \
    } while (0)
    if (p.cfg == 0) {
        if (ni <= 4) QE_F32_LAUNCH(4, 1, 4); else QE_F32_LAUNCH(4, 1, 7);
    } else {
        if (ni <= 4) QE_F32_LAUNCH(2, 2, 2); else QE_F32_LAUNCH(2, 2, 4);
    }
#undef QE_F32_LAUNCH
    QE_LAUNCH_CHECK();
    return QE_OK;
}

}  // namespace qe
