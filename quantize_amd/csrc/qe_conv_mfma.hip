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
#include "qe_conv_mfma_kernel.hpp"

#include <algorithm>
#include <cstdlib>

namespace qe {

struct PrepArgs {
    const uint8_t *w;
    const float *w_scale, *w_zero;
    const float *bias;
    int w_bits, w_sign, w_per_tensor;
    int OC, IC, KK, OCP, NG;
    int KH, KW;
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
    // 2-D prefix table of the per-tap sums: P[i][j] = sum over kh < i, kw < j (border-aware S_w in O(1), epilogue)
    {
        const int PW1 = a.KW + 1, PS = (a.KH + 1) * PW1;
        if (tid < PS) {
            const int i = tid / PW1, j = tid - i * PW1;
            int sum = 0;
            for (int kh = 0; kh < i; ++kh)
                for (int kw = 0; kw < j; ++kw) sum += s_ws[kh * a.KW + kw];
            a.ws[(int64_t)oc * PS + tid] = sum;
        }
    }
    if (tid == 128) {
        float alpha = 0.0f, zwp = 0.0f, b = 0.0f;
        if (live) {
            const float sw = a.w_per_tensor ? a.w_scale[0] : a.w_scale[oc];
            const float zw = a.w_per_tensor ? a.w_zero[0] : a.w_zero[oc];
            alpha = sw;                                   // the epilogue multiplies by the activation scale
            zwp = zw - zero_shift(a.w_bits, a.w_sign);
            b = a.bias ? a.bias[oc] : 0.0f;
        }
        a.ep[oc] = alpha;
        a.ep[a.OCP + oc] = zwp;
        a.ep[2 * a.OCP + oc] = b;
    }
}

// prep for the small-IC kernel: Wt[kh][h][oc][16], byte (kw - 4h)*4 + ic; same ep / ws tables.
__global__ __launch_bounds__(64) void conv_mfma_prep_smallic_kernel(const PrepArgs a, int KH, int KW)
{
    __shared__ int s_ws[64];
    const int oc = blockIdx.x;
    const int tid = threadIdx.x;
    s_ws[tid] = 0;
    __syncthreads();
    const int cb = code_bias(a.w_bits, a.w_sign);
    const bool live = oc < a.OC;
    if (tid < KH * 2) {
        const int kh = tid >> 1, h = tid & 1;
        uint32_t v[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int kw = 4 * h + (j >> 2), ic = j & 3;
            int aw = 0;
            if (live && kw < KW && ic < a.IC) {
                const int64_t e = ((int64_t)oc * a.IC + ic) * a.KK + kh * KW + kw;  // quantconv2d.cu:118
                aw = unpack_code(a.w, e, a.w_bits) - cb;
                atomicAdd(&s_ws[kh * KW + kw], aw);
            }
            v[j >> 2] |= ((uint32_t)aw & 0xffu) << ((j & 3) * 8);
        }
        *reinterpret_cast<uint4 *>(a.wt + (((int64_t)kh * 2 + h) * a.OCP + oc) * 16) = make_uint4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
    {
        const int PW1 = KW + 1, PS = (KH + 1) * PW1;
        for (int e = tid; e < PS; e += 64) {
            const int i = e / PW1, j = e - i * PW1;
            int sum = 0;
            for (int kh = 0; kh < i; ++kh)
                for (int kw = 0; kw < j; ++kw) sum += s_ws[kh * KW + kw];
            a.ws[(int64_t)oc * PS + e] = sum;
        }
    }
    if (tid == 0) {
        float alpha = 0.0f, zwp = 0.0f, b = 0.0f;
        if (live) {
            const float sw = a.w_per_tensor ? a.w_scale[0] : a.w_scale[oc];
            const float zw = a.w_per_tensor ? a.w_zero[0] : a.w_zero[oc];
            alpha = sw;                                   // the epilogue multiplies by the activation scale
            zwp = zw - zero_shift(a.w_bits, a.w_sign);
            b = a.bias ? a.bias[oc] : 0.0f;
        }
        a.ep[oc] = alpha;
        a.ep[a.OCP + oc] = zwp;
        a.ep[2 * a.OCP + oc] = b;
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------

struct MfmaPlan {
    bool ok = false;
    int cfg = 0;       // 0: 4x1 waves (MT 128), 1: 2x2 (MT 64), 2: 1x4 (MT 32)
    int MT = 0, OCP = 0, NCH = 0, NG = 0, KK = 0, OH = 0, OW = 0;
    int TH = 0, ni = 0, niw = 0, IHT = 0, IWP = 0, ROWMUL = 1, COLMUL = 1;
    bool smallic = false;
    int GI = 1, NS = 1;
    bool flat = false, wraw = false, ws = false, s2 = false, sm2 = false;
    bool expand = false;       // sub-8-bit activations are expanded to 8-bit codes in the workspace first
    bool x4 = false;           // 4-bit activations read from the packed stream by the flat kernel itself
    size_t xe_off = 0;
    bool flatg = false;        // flat 1x1 kernel for small planes (several whole images per tile)
    bool sub = false;          // strided 1x1: the sampled pixels are gathered into a dense tensor first
    bool sub_x4 = false;       // ... straight from the 4-bit stream (subsample_x4_kernel), no expansion pass
    size_t sub_off = 0;
    int PADW = 0;
    size_t lds = 0;
    size_t wt_bytes = 0, ep_off = 0, ws_off = 0, total = 0;
    size_t prep_total = 0;     // leading part of the workspace the prep pass fills (x-independent: can be kept across calls)
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// column tiles per wave each layout is instantiated for (descending), and waves along the pixels
static const int kNiw[3][3] = {{7, 4, 2}, {4, 2, 1}, {2, 1, 0}};
static const int kWN[3] = {1, 2, 4};

static MfmaPlan make_plan8(const qe_conv_shape *sh, int x_bits, int w_bits, bool x4 = false)
{
    MfmaPlan p;
    p.OH = (sh->H + 2 * sh->padding - sh->KH) / sh->stride + 1;
    p.OW = (sh->W + 2 * sh->padding - sh->KW) / sh->stride + 1;
    p.KK = sh->KH * sh->KW;
    if (p.OH <= 0 || p.OW <= 0 || sh->N <= 0 || sh->OC <= 0) return p;
    if ((int64_t)sh->IC * sh->H * sh->W >= (1ll << 31)) return p;
    // int32 accumulators: |a_x a_w| <= 2^14 per product, so a reduction of 2^17 or more terms could overflow silently where
    // the reference (fp32 accumulation, quantconv2d.cu:84) merely rounds -> those problems keep the order-preserving fp32 kernel
    if ((int64_t)sh->IC * sh->KH * sh->KW >= (1ll << 17)) return p;
    if ((int64_t)sh->N * sh->IC * sh->H * sh->W < 64) return p;  // clamped 8-byte reads need a stream >= 8 bytes
    if (sh->W < 4) return p;                                       // rows are fetched in 4-pixel quads
    if ((int64_t)sh->OC * p.OH * p.OW >= (1ll << 29)) return p;      // 32-bit store offsets inside one image
    p.cfg = sh->OC > 64 ? 0 : (sh->OC > 32 ? 1 : 2);
    p.MT = p.cfg == 0 ? 128 : (p.cfg == 1 ? 64 : 32);
    // 4x1 waves: 7 column tiles per wave (112 accumulator registers) keeps the 3x3 variant, which
    // also holds 9 weight fragments, inside 256 VGPRs; 224 pixels = 4 rows of 56 / 8 of 28 / 14x14+.
    const int max_tiles = kNiw[p.cfg][0] * kWN[p.cfg];
    if (p.KK > 64 || p.OW > 32 * max_tiles) return p;
    p.OCP = (sh->OC + p.MT - 1) / p.MT * p.MT;
    const int NQ = (sh->W + 3) / 4;
    const int P = sh->H * sh->W;
    p.flat = p.KK == 1 && sh->stride == 1 && sh->padding == 0 && x_bits == 8 && (P % 4) == 0 && P >= 64 && sh->IC >= 16;
    // 1x1 / stride 2 / no padding (the downsample branches): same GEMM over the flat OUTPUT pixels, the
    // staging keeps the even columns of the even input rows.  224-pixel tiles must hold whole output rows.
    const int POUT = p.OH * p.OW;
    if (!p.flat && p.KK == 1 && sh->stride == 2 && sh->padding == 0 && x_bits == 8 && p.cfg == 0 && sh->IC >= 64 &&
        (POUT % 4) == 0 && POUT >= 64 && 224 % p.OW == 0 && sh->W >= 16 && (sh->W % 4) == 0 &&
        !(env_get("QE_FLAT_S2") && atoi(env_get("QE_FLAT_S2")) == 0)) {
        const int rt = 224 / p.OW, seg = (sh->W + 15) / 16;
        if (64 * rt * seg <= 8 * MF_THREADS) { p.flat = true; p.s2 = true; }
    }
    // 1x1 / stride 1 / no padding on 49..56-pixel planes (7x7 maps): the flat kernel's small-plane variant
    // (conv_mfma_flatg_kernel).  QE_FLATG=0 leaves these layers on the halo kernel.
    if (!p.flat && p.KK == 1 && sh->stride == 1 && sh->padding == 0 && x_bits == 8 && p.cfg == 0 && sh->IC >= 64 &&
        (P + 7) / 8 == 7 && (int64_t)sh->N * sh->IC * P < (1ll << 32) &&
        !(env_get("QE_FLATG") && atoi(env_get("QE_FLATG")) == 0)) {
        const int nch = (sh->IC + 31) / 32;
        p.flatg = true;
        p.IWP = 56;                                   // slots per image (P rounded up to 8)
        p.GI = std::max(1, std::min((int)sh->N, 224 / p.IWP));
        p.NS = nch >= 4 ? 4 : 2;
        p.lds = std::max((size_t)(32 * p.NS) * 224, (size_t)4 * 32 * 36 * 4) + (size_t)224 * 4;
        p.TH = 1; p.ni = 7; p.niw = 7;
        p.NCH = nch; p.NG = 2 * nch;
        p.wraw = (w_bits == 8) && (sh->IC % 16) == 0;
        p.wt_bytes = p.wraw ? 0 : (size_t)p.NG * p.OCP * 16;
        p.IHT = 1;
    }
    if (p.flatg) {
    } else if (p.flat && p.s2) {
        const int ntp = 224, rstr = 224;
        p.NS = 2;
        p.lds = std::max((size_t)64 * rstr, (size_t)4 * 32 * 36 * 4) + (size_t)ntp * 4;
        p.TH = 1; p.ni = 7; p.niw = 7;
        p.NCH = (sh->IC + 31) / 32; p.NG = 2 * p.NCH;
        p.wraw = (w_bits == 8) && (sh->IC % 16) == 0;
        p.wt_bytes = p.wraw ? 0 : (size_t)p.NG * p.OCP * 16;
        p.IHT = (POUT + ntp - 1) / ntp;   // pixel tiles per image
    } else if (p.flat) {
        // 1x1 / stride 1 / no padding: GEMM over the flat pixel index (conv_mfma_flat_kernel)
        int tiles = max_tiles;
        if (p.cfg == 0) {
            // shallow layers (a single stage) are latency- not MFMA-bound: 128-pixel tiles keep the
            // accumulators small enough for a third workgroup per CU.  Then tile quantisation: a plane of
            // 784 pixels (28x28) wastes 12.5 % of 224- or 128-pixel tiles but only 2 % of 160-pixel ones,
            // so the width with clearly less padding wins.  QE_FLAT_NIW overrides (tuning).
            const char *ov = env_get("QE_FLAT_NIW");
            const int forced = ov ? atoi(ov) : 0;
            if (forced == 4 || forced == 5 || forced == 7) tiles = forced;
            else {
                auto waste = [&](int t) { return (double)((P + 32 * t - 1) / (32 * t)) * (32 * t) / (double)P; };
                tiles = sh->IC <= 128 ? 4 : 7;
                static const int cands[3] = {7, 5, 4};
                for (int c : cands)
                    if (waste(c) < waste(tiles) - 0.03) tiles = c;
                // measured exceptions (profiles/r02y_ab_flat_niw.txt, cold per-layer A/B on ResNet-50 at batch 256):
                //   128 -> 512 @28x28: 128-pixel tiles (a fourth workgroup per CU) beat the better-fitting 160-pixel ones by 5 %;
                //   512 -> 128 @28x28: 224-pixel tiles beat 160-pixel ones by 4.5 % (1024 workgroups = two full rounds).
                if (P == 784 && sh->IC <= 128 && sh->OC >= 256) tiles = 4;
                if (P == 784 && sh->IC >= 512 && sh->OC <= 128) tiles = 7;
            }
        }
        const int ntp = 32 * tiles;
        const int rstr = 32 * (tiles | 1);
        const int nch = (sh->IC + 31) / 32;
        p.NS = 1;
        int ns_max = 4;
        if (const char *e = env_get("QE_FLAT_NS")) ns_max = std::max(1, atoi(e));   // tuning knob
        // 64 -> 256 @56x56 (write-bound, two chunks in all): one chunk per stage is 3 % faster (r02y_ab_flat_ns.txt)
        if (!env_get("QE_FLAT_NS") && nch == 2 && sh->OC >= 4 * sh->IC && P >= 3136) ns_max = 1;
        for (int cand = 4; cand > 1; cand >>= 1)
            if (cand <= ns_max && cand <= nch && (size_t)(32 * cand) * rstr + (size_t)ntp * 4 <= (size_t)MF_MAX_LDS) { p.NS = cand; break; }
        p.lds = std::max((size_t)(32 * p.NS) * rstr, (size_t)4 * 32 * 36 * 4) + (size_t)ntp * 4;
        p.TH = 1; p.ni = tiles; p.niw = tiles / kWN[p.cfg];
        p.NCH = nch; p.NG = 2 * nch;
        p.wraw = (w_bits == 8) && (sh->IC % 16) == 0 && !x4;   // the 4-bit-activation instances take prepared fragments only
        p.wt_bytes = p.wraw ? 0 : (size_t)p.NG * p.OCP * 16;
        p.IHT = (P + ntp - 1) / ntp;   // pixel tiles per image
    } else
    p.smallic = sh->IC <= 4 && sh->KW <= 8 && sh->KH <= 8 && x_bits == 8;
    // 3x3, 8-bit activations, more than 32 output channels: two strips per wave, weights through LDS
    // (conv_mfma_sm2_kernel).  Measured against the halo / warp-specialised kernels on ResNet-50 (tools/ab_env.sh
    // QE_SM2 0 1): 56x56 64->64 0.083 -> 0.068 ms, 14x14 256->256 0.050 -> 0.048, 28x28 +4 %, 7x7 maps and the
    // stride-2 layers +15 % (the warp-specialised kernel / bigger halo tiles win there).  Default: stride 1 and a
    // tile that is either 64 channels wide or a whole image; QE_SM2=1 forces it wherever it fits, QE_SM2=0 never.
    const int sm2_env = env_get("QE_SM2") ? atoi(env_get("QE_SM2")) : -1;
    if (!p.flat && !p.flatg && !p.smallic && p.KK == 9 && sh->KW == 3 && sh->KH == 3 && x_bits == 8 && p.cfg <= 1 && sm2_env != 0) {
        const int max_px = 32 * (p.cfg == 0 ? 8 : 16);
        int GI = 1;
        if (p.OH * p.OW <= max_px / 2) GI = std::max(1, std::min((int)sh->N, max_px / (p.OH * p.OW)));
        int TH = (GI > 1) ? p.OH : std::min(p.OH, max_px / p.OW);
        if (GI == 1 && TH >= 1) { const int nt = (p.OH + TH - 1) / TH; TH = (p.OH + nt - 1) / nt; }   // balanced row tiles
        while (TH >= 1) {
            const int IHT = (TH - 1) * sh->stride + 3, IWP = (p.OW - 1) * sh->stride + 3;
            const int units = GI * IHT * NQ;
            const size_t gsz = (size_t)GI * IHT * IWP;
            const size_t wpieces = ((size_t)9 * 2 * p.MT + MF_THREADS - 1) / MF_THREADS * MF_THREADS;   // whole piece rounds
            const size_t lds = align_up((2 * gsz + MF_TRASH) * 16 + gsz * 4, 16) + wpieces * 16;
            if (units <= MF_THREADS && lds <= (size_t)MF_MAX_LDS_SM2) {
                p.sm2 = true; p.GI = GI; p.TH = TH; p.IHT = IHT; p.IWP = IWP; p.lds = lds; p.NS = 1;
                break;
            }
            if (GI > 1) { --GI; continue; }
            --TH;
        }
        if (p.sm2) {
            p.NCH = (sh->IC + 31) / 32;
            p.NG = 2 * p.NCH;
            p.ni = (p.GI * p.TH * p.OW + 31) / 32;
            p.niw = 4;
            p.wt_bytes = (size_t)p.KK * p.NG * p.OCP * 16;
            if ((int64_t)p.wt_bytes >= (1ll << 31)) p.sm2 = false;
            if (sm2_env < 0 && !(sh->stride == 1 && p.GI == 1 && (p.cfg == 1 || p.TH == p.OH))) p.sm2 = false;
            if (!p.sm2) { p.GI = 1; p.TH = 0; p.NS = 1; }   // the halo plan below starts from scratch
        }
    }
    if (p.flat || p.flatg || p.sm2) {
    } else if (p.smallic) {
        // stem layout: K = (kh) x [kw 0..7][ic 0..3]; the whole (tiny) channel depth is one stage
        p.NCH = 1;
        p.NG = 2;
        p.niw = kNiw[p.cfg][0];
        int stem_tiles = max_tiles;
        // 64-channel workgroups (the ResNet stem): 7 column tiles per wave = 4 output rows per tile instead of 2
        // (fewer, larger workgroups: less halo re-read, prologue amortised).  QE_STEM_NIW=4 restores the old tiles.
        if (p.cfg == 1 && !(env_get("QE_STEM_NIW") && atoi(env_get("QE_STEM_NIW")) == 4) && p.OW <= 32 * 14) {
            p.niw = 7;
            stem_tiles = 14;
        }
        int TH = std::min(p.OH, (32 * stem_tiles) / p.OW);
        for (; TH >= 1; --TH) {
            const int IHT = (TH - 1) * sh->stride + sh->KH;
            const int IWP = (p.OW - 1) * sh->stride + 8;
            const size_t lds = ((size_t)IHT * IWP * 2 + MF_TRASH) * 4;
            if (lds <= (size_t)MF_MAX_LDS) { p.TH = TH; p.IHT = IHT; p.IWP = IWP; p.lds = lds; break; }
        }
        if (p.TH == 0) return p;
        p.ni = (p.TH * p.OW + 31) / 32;
        p.wt_bytes = (size_t)sh->KH * 2 * p.OCP * 16;
    } else {
        p.NCH = (sh->IC + 31) / 32;
        p.ROWMUL = (sh->KH == 1) ? sh->stride : 1;   // 1xK strided: only every stride-th row is ever read
        p.COLMUL = (sh->KW == 1) ? sh->stride : 1;
        const int max_px = 32 * max_tiles;
        // small feature maps (7x7): several whole images per tile, so a weight fragment and a
        // barrier pair are amortised over 7 column tiles instead of 2
        if (p.OH * p.OW <= max_px / 2) p.GI = std::max(1, std::min((int)sh->N, max_px / (p.OH * p.OW)));
        int TH = (p.GI > 1) ? p.OH : std::min(p.OH, max_px / p.OW);
        for (;;) {
            const int IHT = (p.ROWMUL > 1) ? TH : (TH - 1) * sh->stride + sh->KH;
            const int IWP = (p.COLMUL > 1) ? p.OW : (p.OW - 1) * sh->stride + sh->KW;
            const int units = p.GI * IHT * NQ;
            // chunks per stage: as many as the idle staging threads and LDS allow (1x1, 8-bit only)
            int ns = 1;
            if (p.KK == 1 && x_bits == 8) {
                for (int cand = 4; cand > 1; cand >>= 1) {
                    const size_t l = ((size_t)2 * cand * p.GI * IHT * IWP + MF_TRASH) * 16 + (size_t)p.GI * IHT * IWP * 4;
                    if (cand <= p.NCH && units * cand <= MF_THREADS && l <= (size_t)MF_MAX_LDS) { ns = cand; break; }
                }
            }
            const size_t lds = ((size_t)2 * ns * p.GI * IHT * IWP + MF_TRASH) * 16 + (size_t)p.GI * IHT * IWP * 4;
            if (lds <= (size_t)MF_MAX_LDS && units <= MF_THREADS) {
                p.TH = TH; p.IHT = IHT; p.IWP = IWP; p.lds = lds; p.NS = ns;
                break;
            }
            if (p.GI > 1) { --p.GI; continue; }
            if (--TH < 1) break;
        }
        if (p.TH == 0) return p;
        p.NCH = (p.NCH + p.NS - 1) / p.NS * p.NS;   // padded chunks carry zero weights
        p.NG = 2 * p.NCH;
        p.ni = (p.GI * p.TH * p.OW + 31) / 32;
        p.niw = kNiw[p.cfg][0];
        for (int i = 0; i < 3; ++i)
            if (kNiw[p.cfg][i] > 0 && kNiw[p.cfg][i] * kWN[p.cfg] >= p.ni) p.niw = kNiw[p.cfg][i];
        p.wt_bytes = (size_t)p.KK * p.NG * p.OCP * 16;
        // 3x3, 8-bit activations, 128-channel tiles: the warp-specialised kernel (producer/consumer
        // waves, double-buffered halo image).
        const char *ws_env = env_get("QE_WS");
        // Measured on ResNet-50 (A/B, tools/ab_env.sh QE_WS): it wins where a workgroup has little MFMA work
        // per stage to hide its own fetch behind (7x7 maps: 0.068 -> 0.052-0.057 ms) and loses 5-15 % on the
        // 14x14 / 28x28 / 56x56 layers, where two resident single-role workgroups overlap each other better
        // than one specialised one (stamps: the consumer issues one MFMA per ~60 cycles; its weight loads queue
        // behind the producers' HBM misses in the CU's in-order vector-memory path).  QE_WS=1 forces it on.
        const bool ws_default = p.GI > 1 || p.OH * p.OW <= 64;
        const bool ws_on = ws_env ? atoi(ws_env) != 0 : ws_default;
        if (p.KK == 9 && sh->KW == 3 && x_bits == 8 && p.cfg == 0 && p.NS == 1 && ws_on) {
            // stride 1 with padding 1: unpadded LDS rows (conflict-free fragment reads) + lane masks
            const char *np_env = env_get("QE_WS_NOPAD");
            const bool nopad = sh->stride == 1 && sh->padding == 1 && (np_env && atoi(np_env) == 1);   // off by default
            const int iwp = nopad ? sh->W : p.IWP;
            const int gd = nopad ? sh->padding : 0;
            const size_t gsz = (size_t)p.GI * p.IHT * iwp + 2 * gd;
            const size_t lds = ((size_t)4 * gsz + MF_TRASH) * 16 + gsz * 4;
            if (lds <= (size_t)MF_MAX_LDS) { p.ws = true; p.lds = lds; p.IWP = iwp; p.PADW = nopad ? 0 : sh->padding; }
        }
    }
    if ((p.flat || p.flatg) && p.wraw) { p.total = 0; p.ok = true; return p; }
    p.ep_off = align_up(p.wt_bytes, 256);
    p.ws_off = align_up(p.ep_off + (size_t)3 * p.OCP * sizeof(float), 256);
    p.total = align_up(p.ws_off + (size_t)p.OCP * (sh->KH + 1) * (sh->KW + 1) * sizeof(int), 256);
    p.ok = true;
    return p;
}

// Sub-8-bit activations: the halo kernel can decode them on the fly (8-byte clamped reads + shifts per 4 pixels),
// but that path is 3-4x slower than the 8-bit kernels (ResNet-50 W4A4: 16.0 ms vs 4.7 ms per batch-256).  Instead
// the stream is expanded once to signed 8-bit stored codes in the workspace (one pass at HBM rate: b/8 + 1 bytes per
// element) and every fast 8-bit kernel applies.  QE_EXPAND=0 keeps the in-kernel decode (tuning / tests).
// the dense problem a strided 1x1 / pad 0 convolution reduces to: out[n,oc,oh,ow] only ever reads x[n,c,oh*s,ow*s]
static qe_conv_shape dense_shape(const qe_conv_shape *sh)
{
    qe_conv_shape d = *sh;
    d.H = (sh->H - 1) / sh->stride + 1;
    d.W = (sh->W - 1) / sh->stride + 1;
    d.stride = 1;
    return d;
}

// Stride-2 gather without index divisions: a plane's OH x nq units (nq = 16-byte pieces per even input row, a power of two)
// sit in UP = 2^LOG_UP consecutive threads, thread u -> row u / nq, piece u % nq by shifts; a block takes 256 / UP planes per
// round and 4 rounds, all 4 loads of a thread issued before its stores.  One unaligned 16-byte load (it may run into the
// following odd row, never past the tensor: 16 nq <= 2 W, H even), two v_perm keep the even bytes, one 8-byte store (the last
// piece of a row in 4 / 2 / 1-byte steps).  The round-1 kernel spent ~3 integer divisions per unit and took byte gathers
// on 14-wide rows: 2.5 TB/s over the two ResNet-50 gathers.
template <int LOG_UP>
__global__ __launch_bounds__(256) void subsample2_kernel(const uint8_t *__restrict__ x, uint8_t *__restrict__ y, int64_t n_planes,
                                                         int H, int W, int OH, int OW, int log_nq)
{
    constexpr int UP = 1 << LOG_UP, PPR = 256 / UP, ROUNDS = 4;
    const int u = threadIdx.x & (UP - 1);
    const int oh = u >> log_nq, q = u & ((1 << log_nq) - 1);
    const bool unit_ok = oh < OH && 8 * q < OW;
    const int64_t plane0 = (int64_t)blockIdx.x * (PPR * ROUNDS) + (threadIdx.x >> LOG_UP);
    uint4 d[ROUNDS];
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k) {
        const int64_t pl = plane0 + k * PPR;
        const bool ok = unit_ok && pl < n_planes;
        const uint8_t *src = x + (ok ? (pl * H + 2 * oh) * (int64_t)W + 16 * q : 0);
        __builtin_memcpy(&d[k], src, 16);
    }
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k) {
        const int64_t pl = plane0 + k * PPR;
        if (!(unit_ok && pl < n_planes)) continue;
        const uint32_t lo = __builtin_amdgcn_perm(d[k].y, d[k].x, 0x06040200u);   // even bytes of dwords 0, 1
        const uint32_t hi = __builtin_amdgcn_perm(d[k].w, d[k].z, 0x06040200u);
        uint8_t *dst = y + (pl * OH + oh) * (int64_t)OW + 8 * q;
        const int left = OW - 8 * q;
        if (left >= 8) {
            const uint2 o = make_uint2(lo, hi);
            __builtin_memcpy(dst, &o, 8);
        } else {
            uint32_t v = lo;
            int done = 0;
            if (left >= 4) { __builtin_memcpy(dst, &lo, 4); done = 4; v = hi; }
            if (left - done >= 2) { const uint16_t h2 = (uint16_t)v; __builtin_memcpy(dst + done, &h2, 2); done += 2; v >>= 16; }
            if (left - done >= 1) dst[done] = (uint8_t)v;
        }
    }
}

// 4-bit activations of a stride-2 1x1 layer: out[r][ow] = 8-bit stored code (q + 128) of in[r_in][2 ow], r = (plane, oh),
// r_in = plane H + 2 oh.  One thread per 8 output bytes = 8 input bytes (16 elements, the even ones are the low nibbles):
// one byte-aligned 8-byte load, a mask and an add, one 8-byte store; the last unit of a row goes byte by byte.
// Replaces expand_codes_s8 over the WHOLE tensor followed by subsample_kernel / the in-kernel stride-2 staging.
__global__ __launch_bounds__(256) void subsample_x4_kernel(const uint8_t *__restrict__ x, uint8_t *__restrict__ y, int64_t n_rows,
                                                           int H, int W, int OH, int OW, int sign)
{
    const int nq = (OW + 7) >> 3;
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= n_rows * nq) return;
    const int64_t r = u / nq;
    const int q = (int)(u - r * nq);
    const int64_t plane = r / OH;
    const int oh = (int)(r - plane * OH);
    const uint8_t *src = x + ((plane * H + 2 * oh) * (int64_t)W >> 1) + 8 * q;      // 2 elements per byte: input column 16 q
    uint8_t *dst = y + r * OW + 8 * q;
    const uint64_t add = sign ? 0x7878787878787878ull : 0x8080808080808080ull;     // q + 128 = nibble - 8 + 128 | nibble + 128
    if (8 * q + 8 <= OW) {
        uint64_t v;
        __builtin_memcpy(&v, src, 8);
        v = (v & 0x0f0f0f0f0f0f0f0full) + add;
        __builtin_memcpy(dst, &v, 8);
    } else {
        for (int j = 0; 8 * q + j < OW; ++j) dst[j] = (uint8_t)((src[j] & 0x0f) + (uint8_t)add);
    }
}

static MfmaPlan make_plan(const qe_conv_shape *sh, int x_bits, int w_bits)
{
    const bool expand = x_bits < 8 && !(env_get("QE_EXPAND") && atoi(env_get("QE_EXPAND")) == 0);
    const int xb = expand ? 8 : x_bits;
    // Strided 1x1 (the ResNet downsample branches): gather the sampled pixels once (read every other row, write 1/s^2 of
    // the bytes) and run the stride-1 kernels on the dense tensor, instead of staging 2-4x the needed bytes in every
    // one of the OC/128 workgroups that share a pixel tile.  QE_SUBSAMPLE=0 keeps the in-kernel strided staging.
    // Measured (rocprofv3, in the stack): 512->1024 @28->14 187 -> 137 us, 1024->2048 @14->7 143 -> 109 us; on
    // 256->512 @56->28 the gather (93 us) costs more than it saves, so output planes above 256 pixels keep the flat
    // kernel's in-kernel stride-2 staging.  QE_SUBSAMPLE=1 forces the gather, =0 disables it.
    const int sub_env = env_get("QE_SUBSAMPLE") ? atoi(env_get("QE_SUBSAMPLE")) : -1;
    const int p_out = ((sh->H - 1) / std::max(1, (int)sh->stride) + 1) * ((sh->W - 1) / std::max(1, (int)sh->stride) + 1);
    // 4-bit activations, stride 2: ONE pass reads the even nibbles of the even rows and writes dense 8-bit codes
    // (subsample_x4_kernel) instead of expanding the whole tensor first -- there the gather pays on every plane size
    const bool sub_x4 = x_bits == 4 && expand && sh->stride == 2 && (sh->W % 2) == 0 && ((int64_t)sh->H * sh->W % 2) == 0 &&
                        !(env_get("QE_SUB_X4") && atoi(env_get("QE_SUB_X4")) == 0);
    const bool sub = sh->KH == 1 && sh->KW == 1 && sh->stride > 1 && sh->padding == 0 && xb == 8 && sub_env != 0 &&
                     (sub_env > 0 || p_out <= 256 || sub_x4);
    const qe_conv_shape ds = dense_shape(sh);
    // 4-bit activations on a stride-1 1x1 layer with 128-channel workgroups: the flat kernel unpacks the nibbles in its
    // staging registers (QE_X4=0: expansion pass + 8-bit kernel as for every other sub-8-bit case)
    if (x_bits == 4 && expand && !sub && !(env_get("QE_X4") && atoi(env_get("QE_X4")) == 0)) {
        MfmaPlan q = make_plan8(sh, 8, w_bits, true);
        if (q.ok && q.flat && !q.s2 && !q.flatg && q.cfg == 0) {
            q.x4 = true;
            q.prep_total = q.total;
            return q;
        }
    }
    MfmaPlan p = make_plan8(sub ? &ds : sh, xb, w_bits);
    p.prep_total = p.ok ? p.total : 0;
    if (p.ok && sub) {
        p.sub = true;
        p.sub_x4 = sub_x4;
        p.sub_off = align_up(p.total, 256);
        p.total = p.sub_off + align_up((size_t)ds.N * ds.IC * ds.H * ds.W, 256);
    } else if (sub) {
        p = make_plan8(sh, xb, w_bits);
        p.prep_total = p.ok ? p.total : 0;
    }
    if (p.ok && expand) {
        p.expand = true;
        p.xe_off = align_up(p.total, 256);
        p.total = p.xe_off + align_up((size_t)sh->N * sh->IC * sh->H * sh->W, 256);
    }
    return p;
}

// out[r][ow] = in[r_in][ow * s] for the rows r = (n*IC + c)*OH + oh.
// WIDE (stride 2, W % 4 == 0, W >= 16): one thread per 8 output bytes = one 16-byte load (clamped to end at the row's
// end and rotated back by whole dwords, so nothing is read past a row), two v_perm, one 8-byte store.
// otherwise: one thread per 4 output bytes, byte gathers.
template <bool WIDE>
__global__ __launch_bounds__(256) void subsample_kernel(const uint8_t *__restrict__ x, uint8_t *__restrict__ y, int64_t n_planes,
                                                        int H, int W, int OH, int OW, int s)
{
    constexpr int OPT = WIDE ? 8 : 4;            // output bytes per unit
    constexpr int UPT = 4;                       // units per thread, all loads issued before the first store
    const int nq = (OW + OPT - 1) / OPT;
    const int U = OH * nq;                       // units of one plane
    const int ppb = U >= 256 * UPT ? 1 : (256 * UPT) / U;   // planes per workgroup (32-bit index math only)
    for (int t0 = threadIdx.x; t0 < ppb * U; t0 += 256 * UPT) {
        uint4 d[UPT];
        uint32_t g[UPT];
        uint8_t *dst[UPT];
        int ow0[UPT], rot[UPT];
        bool live[UPT];
#pragma unroll
        for (int k = 0; k < UPT; ++k) {
            const int t = t0 + 256 * k;
            const int pl = t / U, u = t - pl * U;
            const int64_t plane = (int64_t)blockIdx.x * ppb + pl;
            live[k] = t < ppb * U && plane < n_planes;
            const int64_t pc = live[k] ? plane : 0;
            const int oh = u / nq, q = u - oh * nq;
            const uint8_t *src = x + (pc * H + (int64_t)oh * s) * W;
            ow0[k] = OPT * q;
            dst[k] = y + (pc * OH + oh) * OW + ow0[k];
            if constexpr (WIDE) {
                const int iw = 2 * ow0[k];                                // first input column of this unit
                const int iwc = iw < W - 16 ? iw : W - 16;                // 16 bytes that end inside the row
                __builtin_memcpy(&d[k], src + iwc, 16);
                rot[k] = (iw - iwc) >> 2;                                 // whole dwords (W % 4 == 0)
            } else {
                uint32_t v = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = ow0[k] + j < OW ? ow0[k] + j : OW - 1;  // clamped: always a valid byte of the row
                    v |= (uint32_t)src[(int64_t)c * s] << (8 * j);
                }
                g[k] = v;
            }
        }
#pragma unroll
        for (int k = 0; k < UPT; ++k) {
            if (!live[k]) continue;
            if constexpr (WIDE) {
                const uint4 dd = d[k];
                const int r = rot[k];
                const uint32_t d0 = r == 0 ? dd.x : (r == 1 ? dd.y : (r == 2 ? dd.z : dd.w));
                const uint32_t d1 = r == 0 ? dd.y : (r == 1 ? dd.z : (r == 2 ? dd.w : 0u));
                const uint32_t d2 = r == 0 ? dd.z : (r == 1 ? dd.w : 0u);
                const uint32_t d3 = r == 0 ? dd.w : 0u;
                const uint32_t lo = __builtin_amdgcn_perm(d1, d0, 0x06040200u);   // even bytes of d0, d1
                const uint32_t hi = __builtin_amdgcn_perm(d3, d2, 0x06040200u);
                if (ow0[k] + 8 <= OW && (reinterpret_cast<uintptr_t>(dst[k]) & 3) == 0) {
                    const uint2 o = make_uint2(lo, hi);
                    __builtin_memcpy(dst[k], &o, 8);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (ow0[k] + j < OW) dst[k][j] = (uint8_t)((j < 4 ? lo : hi) >> (8 * (j & 3)));
                }
            } else {
                if (ow0[k] + 4 <= OW && (reinterpret_cast<uintptr_t>(dst[k]) & 3) == 0) {
                    *reinterpret_cast<uint32_t *>(dst[k]) = g[k];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (ow0[k] + j < OW) dst[k][j] = (uint8_t)(g[k] >> (8 * j));
                }
            }
        }
    }
}

// diagnostic (-DQE_STAMP) builds: where the kernels drop their per-wave phase sums
unsigned long long *g_mfma_dbg = nullptr;   // also read by qe_linear.hip (diagnostic builds)

int expand_codes_s8(const uint8_t *packed, int64_t n, int n_bits, int sign, uint8_t *out, hipStream_t s);   // qe_tpack.hip
int flatd_variant(const qe_conv_shape *sh, const qe_qparam *x, const qe_qparam *w);                          // qe_conv_flatd.hip
int launch_flatd(const qe_qparam *x, const qe_qparam *w, const float *bias, const qe_conv_shape *sh, float *out, hipStream_t s,
                 const RequantHost *rq = nullptr);
bool flatd_requant_ok(const qe_conv_shape *sh, const qe_qparam *x, const qe_qparam *w, const RequantHost *rq);
bool pwr_eligible(const qe_conv_shape *sh, const qe_qparam *x, const qe_qparam *w, const RequantHost *rq);  // qe_conv_pwr.hip
int launch_pwr(const qe_qparam *x, const qe_qparam *w, const float *bias, const qe_conv_shape *sh, float *out, hipStream_t s,
               const RequantHost *rq);
constexpr int QE_FLATD_DEFAULT = 4;   // 7x7 planes only: -17..-20 % there; the wide variants tie or lose to the register-staged kernels (profiles/r02b_ab_flatd.txt)

bool mfma_conv_eligible(const qe_conv_shape *sh, const qe_qparam *x, const qe_qparam *w)
{
    if (x->n_param != 1) return false;  // per-channel activation scale cannot leave the K sum
    return make_plan(sh, x->n_bits, w->n_bits).ok;
}

size_t mfma_conv_workspace_bytes(const qe_conv_shape *sh, int x_bits, int w_bits)
{
    const MfmaPlan p = make_plan(sh, x_bits, w_bits);
    return p.ok ? p.total : 0;
}

// does this layer's kernel carry the fused re-quantisation epilogue (8-bit codes, per-tensor output scale)?
bool mfma_conv_requant_fused(const qe_conv_shape *sh, const qe_qparam *x, const qe_qparam *w, int rq_bits, int rq_n_param)
{
    if (x->n_param != 1 || rq_bits != 8 || rq_n_param != 1) return false;
    const MfmaPlan p = make_plan(sh, x->n_bits, w->n_bits);
    if (!p.ok) return false;
    if (p.flat || p.flatg) {
        const qe_conv_shape ds = p.sub ? dense_shape(sh) : *sh;
        const size_t patch = p.flatg ? (size_t)p.GI * p.MT * ds.H * ds.W : (size_t)p.MT * 32 * p.ni;
        return align_up(p.lds, 16) + patch <= (size_t)MF_MAX_LDS;
    }
    return true;
}

// bytes of the x-independent part (re-laid-out weights, per-channel constants, tap-sum tables); 0: nothing to prepare
size_t mfma_conv_prepared_bytes(const qe_conv_shape *sh, int x_bits, int w_bits)
{
    const MfmaPlan p = make_plan(sh, x_bits, w_bits);
    return p.ok ? p.prep_total : 0;
}

// What the prepared tables of a problem look like: two problems with the same weights and the same signature share one
// prepared buffer whatever their batch size or image size (0: nothing to prepare).  The prep kernels write
// Wt[tap][NG][OCP][16] (or the stem's per-row layout), 3 x OCP constants and the OCP x (KH+1)(KW+1) prefix table.
uint64_t mfma_conv_prepared_layout(const qe_conv_shape *sh, int x_bits, int w_bits)
{
    const MfmaPlan p = make_plan(sh, x_bits, w_bits);
    if (!p.ok || p.prep_total == 0) return 0;
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { h = (h ^ v) * 1099511628211ull; };
    mix(p.smallic ? 1 : 0); mix((uint64_t)p.OCP); mix((uint64_t)p.NG); mix((uint64_t)p.KK); mix((uint64_t)sh->KH); mix((uint64_t)sh->KW);
    mix((uint64_t)sh->IC); mix((uint64_t)sh->OC); mix((uint64_t)p.prep_total); mix((uint64_t)p.ep_off); mix((uint64_t)p.ws_off);
    return h | 1ull;
}

// mode 0: prepare + run (workspace = [prepared part | scratch]); mode 1: prepare only into `prepared`;
// mode 2: run on a `prepared` buffer filled earlier (workspace = scratch only)
// rq != nullptr: fused re-quantisation -- the epilogues store 8-bit codes into rq->out instead of fp32 into `out`
// (mode 0 or 2).  QE_ERR_UNSUPPORTED when this layer's kernel has no such epilogue (caller: conv + quantize_pack).
int launch_conv_mfma(const qe_qparam *x, const qe_qparam *w, const float *bias, const qe_conv_shape *sh,
                     float *out, void *workspace, size_t workspace_bytes, hipStream_t s, int mode, void *prepared,
                     size_t prepared_bytes, const RequantHost *rq)
{
    const MfmaPlan p = make_plan(sh, x->n_bits, w->n_bits);
    if (!p.ok) return QE_ERR_UNSUPPORTED;
    uint8_t *wsp = static_cast<uint8_t *>(workspace);       // base the plan's offsets are relative to
    uint8_t *prep_base = wsp;
    if (mode == 0) {
        if (p.total > 0) {
            if (workspace == nullptr || workspace_bytes < p.total) return QE_ERR_WORKSPACE;
            if ((reinterpret_cast<uintptr_t>(workspace) & 15) != 0) return QE_ERR_ARG;
        }
    } else {
        if (p.prep_total > 0) {
            if (prepared == nullptr || prepared_bytes < p.prep_total) return QE_ERR_WORKSPACE;
            if ((reinterpret_cast<uintptr_t>(prepared) & 15) != 0) return QE_ERR_ARG;
        }
        prep_base = static_cast<uint8_t *>(prepared);
        if (mode == 2 && p.total > p.prep_total) {
            if (workspace == nullptr || workspace_bytes < p.total - p.prep_total) return QE_ERR_WORKSPACE;
            if ((reinterpret_cast<uintptr_t>(workspace) & 15) != 0) return QE_ERR_ARG;
            wsp = static_cast<uint8_t *>(workspace) - p.prep_total;   // scratch offsets start behind the prepared part
        }
    }
    qe_qparam xe;
    qe_qparam xs;
    qe_conv_shape shd;
    bool sub_done = false;
    if (p.sub && p.sub_x4 && mode != 1) {
        shd = dense_shape(sh);
        const int64_t n_rows = (int64_t)sh->N * sh->IC * shd.H;
        const int64_t units = n_rows * ((shd.W + 7) / 8);
        const int64_t blocks = (units + 255) / 256;
        if (blocks > 0x7fffffffLL) return QE_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(subsample_x4_kernel, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const uint8_t *>(x->data),
                           wsp + p.sub_off, n_rows, (int)sh->H, (int)sh->W, (int)shd.H, (int)shd.W, (int)x->sign);
        QE_LAUNCH_CHECK();
        xs = *x;
        xs.data = wsp + p.sub_off;
        xs.n_bits = 8;
        xs.sign = 1;
        x = &xs;
        sh = &shd;
        sub_done = true;
    }
    if (p.expand && mode != 1 && !sub_done) {
        const int64_t n = (int64_t)sh->N * sh->IC * sh->H * sh->W;
        const int rc = expand_codes_s8(static_cast<const uint8_t *>(x->data), n, x->n_bits, x->sign, wsp + p.xe_off, s);
        if (rc != QE_OK) return rc;
        xe = *x;
        xe.data = wsp + p.xe_off;
        xe.n_bits = 8;
        xe.sign = 1;
        x = &xe;
    }
    if (sub_done) {
    } else if (p.sub && mode == 1) {
        shd = dense_shape(sh);
        sh = &shd;
    } else if (p.sub) {
        shd = dense_shape(sh);
        const int64_t n_planes = (int64_t)sh->N * sh->IC;
        {   // stride 2, even H, a power-of-two number of 16-byte pieces per row that stays inside two input rows: subsample2_kernel
            const int nq = (shd.W + 7) / 8;
            int log_nq = 0;
            while ((1 << log_nq) < nq) ++log_nq;
            const int units2 = shd.H << log_nq;
            if (sh->stride == 2 && (sh->H % 2) == 0 && (1 << log_nq) == nq && 16 * nq <= 2 * sh->W && units2 <= 256 &&
                !(env_get("QE_SUB2") && atoi(env_get("QE_SUB2")) == 0)) {
                int log_up = 3;
                while ((1 << log_up) < units2) ++log_up;
                const int ppb = (256 >> log_up) * 4;
                const int64_t blocks2 = (n_planes + ppb - 1) / ppb;
                if (blocks2 > 0x7fffffffLL) return QE_ERR_UNSUPPORTED;
#define QE_SUB2(L) hipLaunchKernelGGL(subsample2_kernel<L>, dim3((unsigned)blocks2), dim3(256), 0, s, static_cast<const uint8_t *>(x->data), \
                                      wsp + p.sub_off, n_planes, (int)sh->H, (int)sh->W, (int)shd.H, (int)shd.W, log_nq)
                switch (log_up) {
                    case 3: QE_SUB2(3); break; case 4: QE_SUB2(4); break; case 5: QE_SUB2(5); break;
                    case 6: QE_SUB2(6); break; case 7: QE_SUB2(7); break; default: QE_SUB2(8); break;
                }
#undef QE_SUB2
                QE_LAUNCH_CHECK();
                xs = *x;
                xs.data = wsp + p.sub_off;
                x = &xs;
                sh = &shd;
                sub_done = true;
            }
        }
        const bool wide = sh->stride == 2 && (sh->W % 4) == 0 && sh->W >= 16;
        const int opt = wide ? 8 : 4;
        const int units = shd.H * ((shd.W + opt - 1) / opt);
        const int ppb = units >= 1024 ? 1 : 1024 / units;   // 4 units per thread
        const int64_t blocks = (n_planes + ppb - 1) / ppb;
        if (blocks > 0x7fffffffLL) return QE_ERR_UNSUPPORTED;
        if (sub_done) {
        } else if (wide)
            hipLaunchKernelGGL(subsample_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const uint8_t *>(x->data),
                               wsp + p.sub_off, n_planes, (int)sh->H, (int)sh->W, (int)shd.H, (int)shd.W, (int)sh->stride);
        else
            hipLaunchKernelGGL(subsample_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const uint8_t *>(x->data),
                               wsp + p.sub_off, n_planes, (int)sh->H, (int)sh->W, (int)shd.H, (int)shd.W, (int)sh->stride);
        QE_LAUNCH_CHECK();
        xs = *x;
        xs.data = wsp + p.sub_off;
        x = &xs;
        sh = &shd;
    }

    // 1x1 / stride 1 layers with 8-bit operands whose channel depth fits the LDS (IC = 64 | 128 | 256, OC >= 128): the
    // resident-tile kernel (qe_conv_pwr.hip).  QE_PWR=0 keeps the flat kernels below.
    if (pwr_eligible(sh, x, w, rq)) return mode == 1 ? QE_OK : launch_pwr(x, w, bias, sh, out, s, rq);

    // 1x1 / stride 1 layers with 8-bit operands and IC % 64 == 0: the LDS-DMA ring kernel (qe_conv_flatd.hip).
    // QE_FLATD=0 keeps the register-staged flat kernels; QE_FLATD=<bitmask> enables it per tile variant
    // (1: 224-pixel tiles, 2: 160-pixel tiles, 4: 7x7 planes); default from the per-layer A/B in DESIGN.md.
    {
        const int var = flatd_variant(sh, x, w);
        const char *e = env_get("QE_FLATD");
        const int mask = e ? atoi(e) : QE_FLATD_DEFAULT;
        const int bit = var == 7 ? 1 : (var == 5 ? 2 : (var == 8 ? 4 : 0));
        if (var != 0 && (mask & bit) && (rq == nullptr || flatd_requant_ok(sh, x, w, rq))) return mode == 1 ? QE_OK : launch_flatd(x, w, bias, sh, out, s, rq);
    }

    PrepArgs pa;
    pa.w = w->data; pa.w_scale = w->scale; pa.w_zero = w->zero; pa.bias = bias;
    pa.w_bits = w->n_bits; pa.w_sign = w->sign; pa.w_per_tensor = (w->n_param == 1);
    pa.OC = sh->OC; pa.IC = sh->IC; pa.KK = p.KK; pa.OCP = p.OCP; pa.NG = p.NG; pa.KH = sh->KH; pa.KW = sh->KW;
    pa.wt = reinterpret_cast<int8_t *>(prep_base);
    pa.ep = reinterpret_cast<float *>(prep_base + p.ep_off);
    pa.ws = reinterpret_cast<int *>(prep_base + p.ws_off);
    if (((p.flat || p.flatg) && p.wraw) || mode == 2) {
        // nothing to prepare: the kernel reads the packed tensor and builds its constants itself, or the caller kept
        // the prepared tables from an earlier qe_conv_prepare (weights do not change between forward passes)
    } else if (p.smallic)
        hipLaunchKernelGGL(conv_mfma_prep_smallic_kernel, dim3(p.OCP), dim3(64), 0, s, pa, (int)sh->KH, (int)sh->KW);
    else
        hipLaunchKernelGGL(conv_mfma_prep_kernel, dim3(p.OCP), dim3(256), 0, s, pa);
    QE_LAUNCH_CHECK();
    if (mode == 1) return QE_OK;

    MfmaArgs a;
    a.x = x->data;
    a.x_bytes = qe_packed_nbytes((int64_t)sh->N * sh->IC * sh->H * sh->W, x->n_bits);
    a.x_zero = x->zero; a.x_bits = x->n_bits; a.x_sign = x->sign;
    a.wt = pa.wt; a.ep = pa.ep; a.ws = pa.ws; a.out = out;
    a.N = sh->N; a.IC = sh->IC; a.H = sh->H; a.W = sh->W; a.OC = sh->OC; a.KH = sh->KH; a.KW = sh->KW;
    a.stride = sh->stride; a.pad = sh->padding; a.OH = p.OH; a.OW = p.OW;
    a.OCP = p.OCP; a.NG = p.NG; a.NCH = p.NCH;
    a.TH = p.TH; a.tiles_h = (p.OH + p.TH - 1) / p.TH;
    a.GI = p.GI;
    a.PADW = p.ws ? p.PADW : sh->padding;
    a.dbg = g_mfma_dbg;
    a.rq_out = nullptr; a.rq_scale = nullptr; a.rq_zero = nullptr; a.rq_status = nullptr;
    a.rq_qmin = a.rq_qmax = a.rq_lo = a.rq_hi = 0.0f; a.rq_offset = 0; a.rq_patch = 0;
    if (rq != nullptr) {
        if (rq->n_bits != 8 || rq->n_param != 1 || rq->out == nullptr) return QE_ERR_UNSUPPORTED;
        a.rq_out = rq->out; a.rq_scale = rq->scale; a.rq_zero = rq->zero;
        a.rq_qmin = rq->qmin; a.rq_qmax = rq->qmax; a.rq_status = rq->status;
        a.rq_offset = rq->sign ? 128u : 0u;                       // tpack.cu:108-111
        a.rq_lo = rq->sign ? -128.0f : 0.0f; a.rq_hi = rq->sign ? 127.0f : 255.0f;
    }
    a.w_raw = w->data; a.w_scale = w->scale; a.w_zero = w->zero; a.x_scale = x->scale; a.bias = bias;
    a.w_bits = w->n_bits; a.w_sign = w->sign; a.w_per_tensor = (w->n_param == 1);

    a.n_pix_tiles = ((sh->N + p.GI - 1) / p.GI) * a.tiles_h;
    a.n_oc_tiles = p.OCP / p.MT;
    // flat 1x1, deep reductions into >= 256 output channels (1024 -> 256 @14x14, 512 -> 256 @28x28, ...): 8-wave workgroups
    // that own 256 output channels of a pixel tile, so the tile's activations cross the CU's memory path OC / 256 times
    // instead of OC / 128 (these layers are bound by the bytes through that path).  Weights straight
    // from the packed tensor only (no prepared-table layout depends on the channel tile).  Opt-in (QE_FLAT8=1; 3: IC = 1024
    // only; 2: without the deep prefetch): cold per-layer A/B -7 % on 1024 -> 256 @14x14, but the step as a whole does not
    // gain (profiles/r03w_ab_flat8*.txt).
    const bool wide8 = p.flat && !p.s2 && !p.x4 && !p.flatg && p.cfg == 0 && p.wraw && rq == nullptr && p.NS == 4 &&
                       (p.niw == 7 || p.niw == 5) && sh->IC >= 512 && sh->OC % 256 == 0 &&
                       (env_get("QE_FLAT8") ? atoi(env_get("QE_FLAT8")) != 0 && (atoi(env_get("QE_FLAT8")) != 3 || sh->IC == 1024) : false);
    if (wide8) a.n_oc_tiles = sh->OC / 256;
    int64_t n_units = a.n_pix_tiles;             // what the XCD-aware block map distributes
    if (p.flatg) {
        a.tiles_h = 1;                           // one tile = GI whole images
        a.n_pix_tiles = (sh->N + p.GI - 1) / p.GI;
        n_units = a.n_pix_tiles;
    } else if (p.flat) {
        a.tiles_h = p.IHT;                       // pixel tiles per image
        a.n_pix_tiles = sh->N * a.tiles_h;
        n_units = a.n_pix_tiles;   // one pixel tile per workgroup: runs of several tiles with cross-tile
                                   // prefetch were measured and never paid ('what did not work')
    } else {
        n_units = a.n_pix_tiles;
    }
    a.IHT = p.IHT; a.IWP = p.IWP; a.ROWMUL = p.ROWMUL; a.COLMUL = p.COLMUL; a.ni = p.ni;

    // block map: XCD-runs of `chunk` pixel tiles (block_to_tile).  Default: each XCD owns one contiguous
    // eighth of the tiles (sum over the ResNet-50 layers 4.13 -> 4.07 ms against single-tile interleaving);
    // QE_CHUNK_IMAGES = k overrides with runs of k images (0: single tiles).
    {
        const char *ci = env_get("QE_CHUNK_IMAGES");
        const int k = ci ? atoi(ci) : (1 << 20);
        const int64_t per_xcd = (n_units + 7) / 8;
        a.chunk = (int)std::max<int64_t>(1, std::min<int64_t>(per_xcd, (int64_t)k * a.tiles_h));
    }
    const int64_t runs = (n_units + a.chunk - 1) / a.chunk;
    const int64_t groups = (runs + 7) / 8 * a.chunk;
    const int64_t blocks = groups * 8 * a.n_oc_tiles;
    if (blocks > 0x7fffffffLL) return QE_ERR_UNSUPPORTED;
    const bool x8 = x->n_bits == 8;
    // LDS room for the epilogue's copy of the tile's S_w prefix rows (asymmetric activations; stage_ptab)
    size_t lds_e = p.lds;
    a.ptab_off = 0;
    // lane = pixel kernels (halo, sm2, stem) with fused re-quantisation, one image per tile: the codes leave through a
    // workgroup byte patch at the START of the dynamic LDS (<= 32 KB: MT x pixel slots; the staging image is dead by then)
    // instead of as byte stores of 32-byte runs; the epilogue's tables sit behind it.  QE_RQ_PATCH=0: byte stores.
    const bool rq_patch = rq != nullptr && !p.flat && !p.flatg && !p.ws && p.GI == 1 && (p.OH * p.OW) % 4 == 0 &&
                          (p.TH * p.OW) % 4 == 0 && (reinterpret_cast<uintptr_t>(rq->out) & 3) == 0 &&
                          !(env_get("QE_RQ_PATCH") && atoi(env_get("QE_RQ_PATCH")) == 0);
    const size_t stage_bytes = rq_patch ? std::max(p.lds, (size_t)32 * 1024) : p.lds;
    if (rq_patch) { a.rq_patch = 1; lds_e = stage_bytes; }
    if (!p.flat && !p.flatg) {
        const size_t tab = (size_t)p.MT * (sh->KH + 1) * (sh->KW + 1) * sizeof(int);
        const size_t off = align_up(stage_bytes, 16);
        if (off + tab <= (size_t)(p.sm2 ? MF_MAX_LDS_SM2 : MF_MAX_LDS)) { a.ptab_off = (int)off; lds_e = off + tab; }
    }
    // border classes: rows r < n_top have their top taps clipped, the last n_bot rows their bottom taps (columns alike).
    // The class table needs the bands disjoint and (classes) <= (prefix entries per channel) to fit the same LDS slot.
    {
        auto clipped_lo = [](int pad, int stride, int O) { return std::min(O, (pad + stride - 1) / stride); };
        auto clipped_hi = [](int I, int pad, int K, int stride, int O) {
            const int full_last = (I + pad - K) >= 0 ? (I + pad - K) / stride : -1;   // last output index with all taps below the edge
            return std::max(0, std::min(O, O - 1 - full_last));
        };
        a.n_top = clipped_lo(sh->padding, sh->stride, p.OH);
        a.n_bot = clipped_hi(sh->H, sh->padding, sh->KH, sh->stride, p.OH);
        a.n_lft = clipped_lo(sh->padding, sh->stride, p.OW);
        a.n_rgt = clipped_hi(sh->W, sh->padding, sh->KW, sh->stride, p.OW);
        const int ncls = (1 + a.n_top + a.n_bot) * (1 + a.n_lft + a.n_rgt);
        a.ctab = (a.ptab_off != 0 && a.n_top + a.n_bot < p.OH && a.n_lft + a.n_rgt < p.OW &&
                  ncls <= (sh->KH + 1) * (sh->KW + 1) && !(env_get("QE_CTAB") && atoi(env_get("QE_CTAB")) == 0)) ? 1 : 0;
    }
    // flat kernels with fused re-quantisation: room for the workgroup's byte patch behind the staging image
    size_t lds_f = p.lds;
    if (rq != nullptr && (p.flat || p.flatg)) {
        const size_t patch = p.flatg ? (size_t)p.GI * p.MT * sh->H * sh->W : (size_t)p.MT * 32 * p.ni;
        a.ptab_off = (int)align_up(p.lds, 16);
        lds_f = (size_t)a.ptab_off + patch;
        if (lds_f > (size_t)MF_MAX_LDS) return QE_ERR_UNSUPPORTED;
    }
    if (p.flatg) {
        launch_mfma_flatg(a, p.NS, p.wraw, (unsigned)blocks, lds_f, s);
        QE_LAUNCH_CHECK();
        return QE_OK;
    }
    if (p.flat && p.x4) {
        launch_mfma_flat_x4(a, p.niw, p.NS, (unsigned)blocks, lds_f, s);
        QE_LAUNCH_CHECK();
        return QE_OK;
    }
    if (p.flat && wide8) {
        const size_t lds8 = std::max((size_t)(32 * p.NS) * (32 * (p.ni | 1)), (size_t)8 * 32 * 36 * 4) + (size_t)(32 * p.ni) * 4;
        // QE_FLAT8=2: without the deep-prefetch form (one register set of activation pieces, dynamic stage loop)
        const bool deep = (sh->IC == 512 || sh->IC == 1024) && !(env_get("QE_FLAT8") && atoi(env_get("QE_FLAT8")) == 2);
        launch_mfma_flat(a, deep ? 4 : 3, p.niw, p.NS, true, false, (unsigned)blocks, lds8, s);
        QE_LAUNCH_CHECK();
        return QE_OK;
    }
    if (p.flat) {
        launch_mfma_flat(a, p.cfg, p.niw, p.NS, p.wraw, p.s2, (unsigned)blocks, lds_f, s);
        QE_LAUNCH_CHECK();
        return QE_OK;
    }
    if (p.sm2) {
        const int units = p.GI * p.IHT * ((sh->W + 3) / 4);
        const int split = units <= 64 ? 4 : (units <= 128 ? 2 : 1);   // channel slices of the staging threads
        launch_mfma_sm2(a, p.cfg == 0 ? 2 : 1, split, (unsigned)blocks, lds_e, s);
        QE_LAUNCH_CHECK();
        return QE_OK;
    }
    if (p.ws) {
        {
            const int units = p.GI * p.IHT * ((sh->W + 3) / 4);
            const int split = units <= 64 ? 4 : (units <= 128 ? 2 : 1);   // idle producer threads take channel slices
            // (without the class table the ws epilogue reads the prefix rows from global memory: no LDS slot needed)
            launch_mfma_ws(a, p.niw, split, (unsigned)blocks, a.ctab ? lds_e : p.lds, s);
        }
        QE_LAUNCH_CHECK();
        return QE_OK;
    }
    if (p.smallic) {
        launch_mfma_smallic(a, p.cfg, p.niw, (unsigned)blocks, lds_e, s);
        QE_LAUNCH_CHECK();
        return QE_OK;
    }
    switch (p.cfg) {
        case 0: launch_mfma_cfg0(a, p.niw, p.NS, p.KK, x8, (unsigned)blocks, lds_e, s); break;
        case 1: launch_mfma_cfg1(a, p.niw, p.NS, p.KK, x8, (unsigned)blocks, lds_e, s); break;
        default: launch_mfma_cfg2(a, p.niw, p.NS, p.KK, x8, (unsigned)blocks, lds_e, s); break;
    }
    QE_LAUNCH_CHECK();
    return QE_OK;
}

}  // namespace qe

// Not part of the public ABI (absent from include/quant_engine.h): set the stamp buffer of a
// -DQE_STAMP diagnostic build.
extern "C" void qe_debug_set_stamp_buffer(unsigned long long *p) { qe::g_mfma_dbg = p; }
