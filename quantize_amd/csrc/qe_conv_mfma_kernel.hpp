// qe_conv_mfma_kernel.hpp -- the int8 MFMA implicit-GEMM convolution kernel template (gfx950).
// Design notes: qe_conv_mfma.hip.  Everything in the hot loop is straight-line code: loads are
// unconditional (clamped address + mask) because hipcc waits vmcnt(0) after every load it has to
// branch around, which serialises the whole activation fetch into one HBM round trip per dword.
#pragma once
#include "qe_common.h"

#include <type_traits>

namespace qe {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int MF_THREADS = 256;
constexpr int MF_UNITS = 2;        // staging units (16 ch x 4 px) per thread per chunk
constexpr int MF_TRASH = 64;       // per-lane LDS slots that swallow masked-off staging writes
constexpr int MF_MAX_LDS = 64 * 1024;
constexpr int MF_MAX_LDS_SM2 = 80 * 1024;   // sm2 kernel: 2 workgroups per CU x 80 KB = the CU's 160 KB

struct MfmaArgs {
    const uint8_t *x;
    int64_t x_bytes;           // length of the packed activation stream (>= 8)
    const float *x_zero;       // per tensor
    int x_bits, x_sign;
    const int8_t *wt;          // [KK][NG][OCP][16]
    const float *ep;           // [3][OCP]: sw, zw', bias   (activation scale applied in the epilogue: the table is x-independent)
    const int *ws;             // [OCP][(KH+1)*(KW+1)]: 2-D prefix sums over (kh, kw) of sum_ic a_w; last entry = all taps
    float *out;
    int N, IC, H, W, OC, KH, KW, stride, pad, OH, OW;
    int OCP, NG, NCH;          // padded oc, 16-channel groups (even), 32-channel chunks
    int TH, tiles_h, n_pix_tiles, n_oc_tiles;
    int IHT, IWP, ROWMUL, COLMUL, ni;
    int GI;                    // whole images per pixel tile (> 1 only for small feature maps, TH == OH)
    int PADW;                  // ws kernel: left padding columns materialised in LDS (0 = unpadded rows + lane masks)
    int chunk;                 // consecutive pixel tiles one XCD takes before the next XCD's run starts
    int ptab_off;              // byte offset in dynamic LDS of the epilogue's S_w table (0 = none)
    int ctab;                  // 1: that table is the per-(border class, channel) correction (stage_ctab), 0: prefix rows
    int n_top, n_bot, n_lft, n_rgt;   // output rows / columns whose taps are clipped at each image edge
    unsigned long long *dbg;   // diagnostic builds (-DQE_STAMP) only: per-wave phase cycle sums
    // raw operands, used by the flat 1x1 kernel (it builds its epilogue constants itself)
    const uint8_t *w_raw;      // packed OIHW weights as the caller passed them
    const float *w_scale, *w_zero, *x_scale, *bias;
    int w_bits, w_sign, w_per_tensor;
    // fused re-quantisation of the output (qe_quantconv2d_requant): rq_out != nullptr -> the epilogue stores the 8-bit
    // code of round(y / scale - zero).clamp(qmin, qmax) (1 byte per element, NCHW) instead of the fp32 y into `out`
    uint8_t *rq_out;
    const float *rq_scale, *rq_zero;   // per tensor (the consumer's activation quantiser; its conv wants one scale anyway)
    float rq_qmin, rq_qmax, rq_lo, rq_hi;   // clamp of the quantiser; representable range of the code (tpack's range test)
    unsigned rq_offset;                     // stored code = (q + offset) & 0xff (tpack.cu:108-111)
    int32_t *rq_status;                     // bit 0 set when a value fails the range test (NaN, or qmin/qmax outside the code range)
    int rq_patch;                           // lane = pixel kernels (halo, sm2, stem), one image per tile: the tile's codes go through a
                                            // workgroup byte patch [MT][32 NIW WN] at the start of the dynamic LDS and leave as 16-byte row pieces
};

// y -> stored 8-bit code with the arithmetic of the fused quantise+pack kernel (qe_tpack.hip tp_quantize + tp_code):
//   r = rint(y / scale - zero) ; clamp to [qmin, qmax] ; code = (int(r) + offset) & 0xff ; flag when r is NaN or outside the
// code range.  An IEEE division per output element (~10 VALU instructions) made the fused epilogue SLOWER than storing fp32
// (4.84 vs 4.17 ms per step), so the quotient comes from Markstein's sequence on a reciprocal taken once per thread:
//   q0 = y * rcp ; e = fma(-scale, q0, y) ; q = fma(e, rcp, q0)
// which IS the correctly rounded y / scale whenever rcp = RN(1 / scale), the significand of scale is not all ones and
// nothing over- or underflows (Markstein 1990; Cornea et al., "Scientific computing on Itanium", thm. 8.3).  y is first
// clamped to +-B with B / |scale| beyond the clamp bounds, which changes no code and keeps infinities out of the fma;
// tiny quotients (where the sequence could round differently) cannot reach a rounding boundary of q - zero.  Scales
// outside those conditions take the division (`slow`, uniform).  When the status flag comes back set the codes are
// unspecified (the reference raises "out of range" there).
struct RqConst {
    float sc, rcp, nsc, zr, qmin, qmax, offf, B, lo, hi;
    bool slow, chk;
};
template <class Args>                                      // MfmaArgs, PwrArgs: the same rq_* fields
__device__ __forceinline__ RqConst rq_setup(const Args &a)
{
    RqConst c;
    c.sc = a.rq_scale[0];
    c.zr = a.rq_zero[0];
    c.rcp = 1.0f / c.sc;
    c.nsc = -c.sc;
    c.qmin = a.rq_qmin; c.qmax = a.rq_qmax; c.lo = a.rq_lo; c.hi = a.rq_hi;
    c.offf = (float)a.rq_offset;
    const float asc = fabsf(c.sc);
    const float span = fmaxf(fabsf(c.qmin), fabsf(c.qmax)) + fabsf(c.zr) + 2.0f;
    c.B = asc * span * 2.0f;
    c.slow = !(asc >= 0x1p-60f && asc <= 0x1p60f) || (__float_as_uint(c.sc) & 0x7fffffu) == 0x7fffffu || !(span <= 0x1p30f) ||
             !(c.qmin <= c.qmax);
    c.chk = !(c.qmin >= c.lo && c.qmax <= c.hi);          // clamp bounds inside the code range: only NaN can fail the range test
    return c;
}
// returns r + offset as a float (0 .. 255 whenever the range test passes)
__device__ __forceinline__ float rq_value(const RqConst &c, float v, bool &bad)
{
    float r;
    if (c.slow) {
        r = rintf(v / c.sc - c.zr);
        r = (r != r) ? r : fminf(fmaxf(r, c.qmin), c.qmax);
        bad |= !(r >= c.lo && r <= c.hi);
    } else {
        bad |= (v != v);
        const float vc = __builtin_amdgcn_fmed3f(v, -c.B, c.B);
        const float q0 = vc * c.rcp;
        const float e = fmaf(c.nsc, q0, vc);
        const float q = fmaf(e, c.rcp, q0);
        r = __builtin_amdgcn_fmed3f(rintf(q - c.zr), c.qmin, c.qmax);
        if (c.chk) bad |= !(r >= c.lo && r <= c.hi);
    }
    return r + c.offf;
}
// Two values per instruction where the ISA has packed fp32 (v_pk_mul_f32, v_pk_fma_f32, v_pk_add_f32: 7.5 instead of 13 VALU
// instructions per output element; the fused epilogue is VALU-bound: 2.8 G elements per batch-256 ResNet-50 step).  Valid
// when rq_fast_ok(c) and the caller has bounded its own constants so that y is finite and |y| <= 2^52 (rq_bounded): then
// no NaN and no overflow can occur anywhere in the sequence, the +-B clamp of rq_value is the identity wherever it matters
// (beyond B the code is the clamp bound either way) and no range flag can be raised -- the same codes as rq_value.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bool rq_fast_ok(const RqConst &c) { return !c.slow && !c.chk; }
// |alpha| <= 2^10, |cst| <= 2^40, |bias| <= 2^50, |zw'| <= 2^20: y = fma(alpha, acc + cst - zw' S_x, bias) stays below 2^52
__device__ __forceinline__ bool rq_bounded(float alpha, float cst, float bias, float zwp)
{
    return fabsf(alpha) <= 0x1p10f && fabsf(cst) <= 0x1p40f && fabsf(bias) <= 0x1p50f && fabsf(zwp) <= 0x1p20f;
}
__device__ __forceinline__ v2f rq_fast2(const RqConst &c, v2f y)
{
#pragma clang fp contract(off)
    const v2f rcp = {c.rcp, c.rcp}, nsc = {c.nsc, c.nsc}, zr = {c.zr, c.zr}, off = {c.offf, c.offf};
    const v2f q0 = y * rcp;
    const v2f e = __builtin_elementwise_fma(nsc, q0, y);
    const v2f q = __builtin_elementwise_fma(e, rcp, q0);
    const v2f d = q - zr;
    v2f r;
    r.x = __builtin_amdgcn_fmed3f(rintf(d.x), c.qmin, c.qmax);
    r.y = __builtin_amdgcn_fmed3f(rintf(d.y), c.qmin, c.qmax);
    return r + off;
}
template <class Args>
__device__ __forceinline__ void rq_report(const Args &a, bool bad)
{
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && (threadIdx.x & 63) == 0 && a.rq_status != nullptr) atomicOr(a.rq_status, 1);
}

// host-side description of a fused output quantiser (qe_requant of the C ABI + destination)
struct RequantHost {
    uint8_t *out;
    const float *scale, *zero;
    int n_param;
    float qmin, qmax;
    int n_bits, sign;
    int32_t *status;
};

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

// 4 consecutive elements of one channel row -> 4 int8 operands in a dword.  `xi` is the image's
// base (wave-uniform), `off` the element offset inside the image (32-bit), `lim` the last offset at
// which a full-width read still ends inside the stream.  Branch-free: the read is clamped and the
// value shifted back; whatever is shifted in belongs to elements that are never valid pixels.
template <bool X8>
__device__ __forceinline__ uint32_t fetch_quad(const uint8_t *__restrict__ xi, int off, int64_t lim,
                                               int n_bits, int cb)
{
    if constexpr (X8) {
        const int offc = off < (int)lim ? off : (int)lim;
        uint32_t v;
        __builtin_memcpy(&v, xi + (uint32_t)offc, 4);  // one (possibly unaligned) global_load_dword
        v >>= 8 * (off - offc);
        return v ^ 0x80808080u;                        // u - 128 per byte: signed q, or unsigned q - 128
    } else {
        const int64_t bit = (int64_t)off * n_bits;
        const int64_t byte = bit >> 3;
        const int64_t bc = byte < lim ? byte : lim;
        uint64_t v;
        __builtin_memcpy(&v, xi + bc, 8);
        v >>= ((int)(bit & 7) + 8 * (int)(byte - bc));
        const uint32_t mask = (1u << n_bits) - 1u;
        uint32_t r = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t c = (uint32_t)(v >> (j * n_bits)) & mask;
            r |= ((c - (uint32_t)cb) & 0xffu) << (8 * j);
        }
        return r;
    }
}

#ifdef QE_STAMP
// In-kernel stamps (guide section 7): one asm statement, fenced, lgkmcnt(0) inside.  Diagnostic build only.
__device__ __forceinline__ unsigned long long qe_stamp()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define QE_ST(i) do { const unsigned long long _t = qe_stamp(); st[i] += _t - tprev; tprev = _t; } while (0)
#else
#define QE_ST(i) do { } while (0)
#endif

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
// Shared epilogue: D (rows = output channel, cols = pixel on the lane) -> fp32 NCHW.
//   out = bias + alpha * (S_aw - zw' S_x - zx' S_w + N_inb zx' zw')      (see qe_conv_mfma.hip)
// A pixel tile is GI images x th rows x OW columns; column q of the tile is image q / OHWt.
// ---------------------------------------------------------------------------------------------
struct TileGeom {
    int n0;      // first image of the tile
    int oh0;     // first output row
    int OHWt;    // pixels per image inside the tile (th * OW)
    int NT;      // valid columns (images that exist x OHWt)
};

template <int WM, int WN, int NIW, bool RQ, bool PATCH = false>
__device__ __forceinline__ void mfma_epilogue_impl(const MfmaArgs &a, v16i (&acc)[NIW], const int (&sxs)[NIW],
                                                   const bool need_sx, const TileGeom g, const int ot,
                                                   const int wm, const int wn, const int col, const int h, const int KK,
                                                   const int *ptab, const float *ctab)
{
    constexpr int MT = 32 * WM;
    const float zxp = a.x_zero[0] - zero_shift(a.x_bits, a.x_sign);
    const bool need_sw = zxp != 0.0f;
    const int oc_base = ot * MT + wm * 32 + 4 * h;  // + (reg&3) + 8*(reg>>2)
    float al[16], bi[16];
    const float sx = a.x_scale[0];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int oc = oc_base + (r & 3) + 8 * (r >> 2);
        al[r] = sx * a.ep[oc];   // alpha = sx * sw (same product the prep pass used to store)
        bi[r] = a.ep[2 * a.OCP + oc];
    }
    const int OHW = a.OH * a.OW;
    const bool full_oc = (ot + 1) * MT <= a.OC;
    // wave-uniform base: image n0, first channel of the wave's 32-row strip, first row of the tile
    const int64_t out_base = ((int64_t)g.n0 * a.OC + ot * MT + wm * 32) * OHW + (int64_t)g.oh0 * a.OW;
    float *out_w = a.out + (RQ ? 0 : out_base);
    // RQ: one byte per element (lanes 0-31 = 32 consecutive bytes of row dr, lanes 32-63 of row dr + 4)
    uint8_t *out_q = RQ ? a.rq_out + out_base : nullptr;
    bool bad = false;
    RqConst rqc;
    if constexpr (RQ) rqc = rq_setup(a);   // per tensor: wave-uniform constants, no registers per row
    // per-lane element offset of column q: image gi of the tile, pixel rq inside it, rows 4h apart
    uint32_t voff[NIW];
    bool valid[NIW];
#pragma unroll
    for (int t = 0; t < NIW; ++t) {
        const int q = (wn + t * WN) * 32 + col;
        const int gi = (a.GI > 1) ? q / g.OHWt : 0;
        const int rq = q - gi * g.OHWt;
        valid[t] = q < g.NT;
        voff[t] = valid[t] ? (uint32_t)(gi * a.OC + 4 * h) * (uint32_t)OHW + (uint32_t)rq : 0u;
    }

    // store of element (register r -> row dr, column slot t): fp32, or (RQ) its 8-bit code as a byte store (lanes 0-31 = 32
    // consecutive bytes of row dr, lanes 32-63 of row dr + 4).  Two wider forms were built and measured slower: a per-wave
    // LDS patch read back as 16-byte pieces (hipcc spilled 600-900 B per lane) and a 4 x 4 byte transpose across lane quads
    // by DPP + v_perm with dword stores (4.63 against 4.41 ms for the fused stack; 264 B of scratch in the 7-slot kernels).
    // PATCH (RQ only): byte (row wm 32 + 4 h + dr, pixel slot q) of the workgroup's patch; slots past the tile and rows past OC
    // hold garbage that rq_patch_copy_out never reads, so no store below needs a test -- and the epilogue no branch
    extern __shared__ __attribute__((aligned(16))) uint8_t qe_ep_smem[];
    constexpr int PSTR = 32 * NIW * WN;
    uint8_t *prow = qe_ep_smem + (wm * 32 + 4 * h) * PSTR + wn * 32 + col;
    auto put = [&](int dr, int t, unsigned code) __attribute__((always_inline)) {
        if constexpr (PATCH) prow[dr * PSTR + t * (WN * 32)] = (uint8_t)code;
        else (out_q + (int64_t)dr * OHW)[voff[t]] = (uint8_t)code;
    };
    auto emit = [&](int r, int dr, int t, float val) __attribute__((always_inline)) {
        if constexpr (RQ && PATCH) {
            bool b = false;                                // slots past the tile / rows past OC must not raise the range flag
            put(dr, t, (unsigned)rq_value(rqc, val, b));
            bad |= b && valid[t] && (full_oc || oc_base + dr < a.OC);
        } else if constexpr (RQ) put(dr, t, (unsigned)rq_value(rqc, val, bad));
        else (out_w + (int64_t)dr * OHW)[voff[t]] = val;
    };

    if (!need_sx && !need_sw) {
        // symmetric operands: out = bias + alpha * S_aw.  Store address = wave-uniform row base
        // (scalar) + one per-lane 32-bit offset: lanes 0-31 are 32 consecutive pixels of row dr,
        // lanes 32-63 of row dr + 4 -> two full 128-byte lines per store instruction.
        // RQ: two registers (channel rows dr, dr + 1) per packed instruction when no check can fire (rq_fast2)
        bool fast = false;
        if constexpr (RQ) {
            rqc.slow = __builtin_amdgcn_readfirstlane(rqc.slow);
            rqc.chk = __builtin_amdgcn_readfirstlane(rqc.chk);
            bool ok = true;
#pragma unroll
            for (int r = 0; r < 16; ++r) ok = ok && rq_bounded(al[r], 0.0f, bi[r], 0.0f);
            fast = rq_fast_ok(rqc) && __builtin_amdgcn_ballot_w64(!ok) == 0ull;
        }
#pragma unroll
        for (int t = 0; t < NIW; ++t) {
            const int q0 = (wn + t * WN) * 32;
            if (PATCH || (full_oc && q0 + 32 <= g.NT)) {
                if (RQ && fast) {
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const int dr = (r & 3) + 8 * (r >> 2);
                        const v2f y = __builtin_elementwise_fma(v2f{al[r], al[r + 1]}, v2f{(float)acc[t][r], (float)acc[t][r + 1]},
                                                                v2f{bi[r], bi[r + 1]});
                        const v2f c2 = rq_fast2(rqc, y);
                        put(dr, t, (unsigned)c2.x);
                        put(dr + 1, t, (unsigned)c2.y);
                    }
                } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    emit(r, (r & 3) + 8 * (r >> 2), t, fmaf(al[r], (float)acc[t][r], bi[r]));
                }
                }
            } else if (!RQ && full_oc) {       // (fp32 stores only: in the re-quantising instances this form spilled 576 B per lane)
                // ragged column tile, whole channel strip: ONE exec region around the 16 stores.  With a test per store
                // every store is a basic block of its own, and hipcc opens each block behind a branch with s_waitcnt
                // vmcnt(0) in kernels that hold LDS-DMA instructions in a loop (sm2, ws): every store of the tile waited
                // for the acknowledgement of the one before it.
                if (valid[t]) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) emit(r, (r & 3) + 8 * (r >> 2), t, fmaf(al[r], (float)acc[t][r], bi[r]));
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    if (valid[t] && oc_base + dr < a.OC) emit(r, dr, t, fmaf(al[r], (float)acc[t][r], bi[r]));
                }
            }
        }
    } else {
        // asymmetric operands.  S_w of a pixel = sum of the per-tap weight sums over the taps INSIDE the image (the
        // reference skips padded taps, quantconv2d.cu:101).  The in-bounds taps of a pixel form a kh-range x kw-range,
        // so S_w comes from the 2-D prefix table the prep pass leaves in a.ws ([oc][(KH+1) x (KW+1)]): interior
        // pixels use the per-channel total (one load per channel, hoisted out of the tile loop), border pixels four
        // table entries.  No per-tap loop, no per-element table walk (the first version cost 3-6x on 3x3 layers and
        // 17x on the 7x7 stem).
        if (ctab != nullptr) {
            // One table lookup per element: T[class][oc] = -zx' S_w(oc, class) + N_inb(class) IC zx' zw'[oc], the class of
            // a pixel being (which clipped row band, which clipped column band) -- no branch between interior and border.
            float zwc[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) zwc[r] = need_sx ? a.ep[a.OCP + oc_base + (r & 3) + 8 * (r >> 2)] : 0.0f;
            const int ncc = 1 + a.n_lft + a.n_rgt;
            const float *crow = ctab + (oc_base - ot * MT);
#pragma unroll
            for (int t = 0; t < NIW; ++t) {
                const int q = (wn + t * WN) * 32 + col;
                const int gi = valid[t] ? q / g.OHWt : 0;
                const int rq = valid[t] ? q - gi * g.OHWt : 0;
                const int rr = rq / a.OW, c = rq - rr * a.OW;
                const int ra = g.oh0 + rr;
                const int rid = ra < a.n_top ? 1 + ra : (ra >= a.OH - a.n_bot ? 1 + a.n_top + (a.OH - 1 - ra) : 0);
                const int cid = c < a.n_lft ? 1 + c : (c >= a.OW - a.n_rgt ? 1 + a.n_lft + (a.OW - 1 - c) : 0);
                const float *ct = crow + (rid * ncc + cid) * MT;
                const int q0 = (wn + t * WN) * 32;
                if (PATCH || (full_oc && q0 + 32 <= g.NT)) {      // wave-uniform: plain stores, no per-element exec masks
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dr = (r & 3) + 8 * (r >> 2);
                        float v = (float)acc[t][r] + ct[dr];
                        if (need_sx) v = fmaf(-zwc[r], (float)sxs[t], v);
                        emit(r, dr, t, fmaf(al[r], v, bi[r]));
                    }
                } else if (!RQ && full_oc) {
                    if (valid[t]) {                    // one exec region (see the symmetric form)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int dr = (r & 3) + 8 * (r >> 2);
                            float v = (float)acc[t][r] + ct[dr];
                            if (need_sx) v = fmaf(-zwc[r], (float)sxs[t], v);
                            emit(r, dr, t, fmaf(al[r], v, bi[r]));
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dr = (r & 3) + 8 * (r >> 2);
                        float v = (float)acc[t][r] + ct[dr];
                        if (need_sx) v = fmaf(-zwc[r], (float)sxs[t], v);
                        const float res = fmaf(al[r], v, bi[r]);
                        if (valid[t] && oc_base + dr < a.OC) emit(r, dr, t, res);
                    }
                }
                }
            if constexpr (RQ) rq_report(a, bad);
            return;
        }
        float zw[16];
        int swt[16];
        const int PW1 = a.KW + 1, PS = (a.KH + 1) * PW1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int oc = oc_base + (r & 3) + 8 * (r >> 2);
            zw[r] = a.ep[a.OCP + oc];
            swt[r] = !need_sw ? 0 : (ptab ? ptab[(oc - ot * MT) * PS + PS - 1] : a.ws[(int64_t)oc * PS + PS - 1]);
        }
#pragma unroll
        for (int t = 0; t < NIW; ++t) {
            const int q = (wn + t * WN) * 32 + col;
            int kh_lo = 0, kh_hi = a.KH, kw_lo = 0, kw_hi = a.KW;
            {
                const int gi = valid[t] ? q / g.OHWt : 0;
                const int rq = valid[t] ? q - gi * g.OHWt : 0;
                const int r = rq / a.OW, c = rq - r * a.OW;
                const int ihb = (g.oh0 + r) * a.stride - a.pad, iwb = c * a.stride - a.pad;
                kh_lo = max(0, -ihb); kh_hi = max(kh_lo, min(a.KH, a.H - ihb));
                kw_lo = max(0, -iwb); kw_hi = max(kw_lo, min(a.KW, a.W - iwb));
            }
            const int n_inb = (kh_hi - kh_lo) * (kw_hi - kw_lo);
            const bool interior = n_inb == KK;
            const float fn = (float)(n_inb * a.IC);
            const int i11 = kh_hi * PW1 + kw_hi, i01 = kh_lo * PW1 + kw_hi, i10 = kh_hi * PW1 + kw_lo, i00 = kh_lo * PW1 + kw_lo;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = (r & 3) + 8 * (r >> 2);
                const int oc = oc_base + dr;
                float v = (float)acc[t][r];
                v = fmaf(-zw[r], (float)sxs[t], v);
                if (need_sw) {
                    int sw_sum = swt[r];
                    if (!interior) {
                        const int *P = ptab ? ptab + (oc - ot * MT) * PS : a.ws + (int64_t)oc * PS;
                        sw_sum = P[i11] - P[i01] - P[i10] + P[i00];
                    }
                    v = fmaf(-zxp, (float)sw_sum, v);
                    v = fmaf(fn * zxp, zw[r], v);
                }
                const float res = fmaf(al[r], v, bi[r]);
                if (PATCH || (valid[t] && oc < a.OC)) emit(r, dr, t, res);
            }
        }
    }
    if constexpr (RQ) rq_report(a, bad);
}

// The patch of the PATCH form -> global memory: [MT][PSTR] codes, row = output channel ot MT + row, slot = pixel of the tile
// (one image per tile: a channel's NT pixels are contiguous from row oh0).  16-byte pieces at dword alignment (host: OH OW, TH OW
// multiples of 4); called by every thread of the workgroup after its waves' epilogues.
template <int MT, int PSTR, int THREADS>
__device__ __forceinline__ void rq_patch_copy_out(const MfmaArgs &a, const TileGeom &g, const int ot, const int tid)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t qe_ep_smem[];
    __syncthreads();
    constexpr int PPR = PSTR / 16;
    const int OHW = a.OH * a.OW;
    const int NT = g.NT;
    for (int e = tid; e < MT * PPR; e += THREADS) {
        const int row = e / PPR, px = 16 * (e - row * PPR);
        const int oc = ot * MT + row;
        if (oc < a.OC && px < NT) {
            const uint4 d4 = *reinterpret_cast<const uint4 *>(qe_ep_smem + row * PSTR + px);
            uint8_t *dst = a.rq_out + ((int64_t)g.n0 * a.OC + oc) * OHW + (int64_t)g.oh0 * a.OW + px;
            if (px + 16 <= NT) {
                __builtin_memcpy(dst, &d4, 16);
            } else {                                                       // NT % 4 == 0: whole dwords
                const uint32_t dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (px + 4 * k < NT) __builtin_memcpy(dst + 4 * k, &dd[k], 4);
            }
        }
    }
}

// RQ is a template parameter of the KERNELS that share this epilogue (own instantiations for the fused re-quantisation:
// as a run-time branch inside one kernel its extra live state pushed the 7-column-tile kernels past 256 VGPRs -- scratch in
// kernels the fp32 path launches).
// PATCH is a template parameter of the KERNELS too (own instantiations, chosen by the launchers from a.rq_patch): as a run-time
// branch between two copies of the epilogue it cost the 7-column-tile kernels 32-576 bytes of scratch and the fused stack
// went 3.5 -> 6.2 ms.
template <int WM, int WN, int NIW, bool RQ = false, bool PATCH = false>
__device__ __forceinline__ void mfma_epilogue(const MfmaArgs &a, v16i (&acc)[NIW], const int (&sxs)[NIW],
                                              const bool need_sx, const TileGeom g, const int ot,
                                              const int wm, const int wn, const int col, const int h, const int KK,
                                              const int *ptab = nullptr,   // LDS copy of this tile's rows of a.ws, or null
                                              const float *ctab = nullptr) // LDS [border class][MT] correction table, or null
{
    mfma_epilogue_impl<WM, WN, NIW, RQ, RQ && PATCH>(a, acc, sxs, need_sx, g, ot, wm, wn, col, h, KK, ptab, ctab);
}

// Asymmetric activations (zx' != 0): the border-aware S_w lookups of the epilogue go to an LDS copy of this tile's MT
// rows of the prefix table instead of global memory (a.ptab_off = byte offset of the copy in dynamic LDS, 0 = none).
// Workgroup-uniform; contains a barrier.
template <int MT>
__device__ __forceinline__ const int *stage_ptab(const MfmaArgs &a, uint8_t *smem_base, int ot, int tid, int nthreads)
{
    const float zxp = a.x_zero[0] - zero_shift(a.x_bits, a.x_sign);
    if (a.ptab_off == 0 || zxp == 0.0f) return nullptr;
    const int PS = (a.KH + 1) * (a.KW + 1);
    int *t = reinterpret_cast<int *>(smem_base + a.ptab_off);
    const int *src = a.ws + (int64_t)ot * MT * PS;
    for (int i = tid; i < MT * PS; i += nthreads) t[i] = src[i];
    __syncthreads();
    return t;
}

// The class form of the same correction (a.ctab): band 0 = no clipping, 1..n_top = output rows 0..n_top-1 (top taps
// clipped), then the last n_bot rows from the bottom up; columns alike.  Built per workgroup from the global prefix
// table: (1 + n_top + n_bot)(1 + n_lft + n_rgt) x MT floats in the LDS slot at a.ptab_off.  Contains a barrier.
template <int MT>
__device__ __forceinline__ const float *stage_ctab(const MfmaArgs &a, uint8_t *smem_base, int ot, int tid, int nthreads)
{
    const float zxp = a.x_zero[0] - zero_shift(a.x_bits, a.x_sign);
    if (a.ptab_off == 0 || a.ctab == 0 || zxp == 0.0f) return nullptr;
    const int PW1 = a.KW + 1, PS = (a.KH + 1) * PW1;
    const int nrc = 1 + a.n_top + a.n_bot, ncc = 1 + a.n_lft + a.n_rgt;
    float *t = reinterpret_cast<float *>(smem_base + a.ptab_off);
    for (int i = tid; i < nrc * ncc * MT; i += nthreads) {
        const int cls = i / MT, ocl = i - cls * MT;
        const int rid = cls / ncc, cid = cls - rid * ncc;
        const int rrep = rid == 0 ? a.n_top : (rid <= a.n_top ? rid - 1 : a.OH - 1 - (rid - 1 - a.n_top));
        const int crep = cid == 0 ? a.n_lft : (cid <= a.n_lft ? cid - 1 : a.OW - 1 - (cid - 1 - a.n_lft));
        const int ihb = rrep * a.stride - a.pad, iwb = crep * a.stride - a.pad;
        const int kh_lo = max(0, -ihb), kh_hi = max(kh_lo, min(a.KH, a.H - ihb));
        const int kw_lo = max(0, -iwb), kw_hi = max(kw_lo, min(a.KW, a.W - iwb));
        const int oc = ot * MT + ocl;
        const int *P = a.ws + (int64_t)oc * PS;
        const int sw_sum = P[kh_hi * PW1 + kw_hi] - P[kh_lo * PW1 + kw_hi] - P[kh_hi * PW1 + kw_lo] + P[kh_lo * PW1 + kw_lo];
        const float fn = (float)((kh_hi - kh_lo) * (kw_hi - kw_lo) * a.IC);
        t[i] = fmaf(fn * zxp, a.ep[a.OCP + oc], -zxp * (float)sw_sum);
    }
    __syncthreads();
    return t;
}

// tile decode shared by both kernels.  XCD-aware block map: blocks b and b+8 share an XCD (and its
// L2); the oc-tiles of one pixel tile get ids that differ by multiples of 8 so they read the same
// activations from one L2.
__device__ __forceinline__ void block_to_tile(const MfmaArgs &a, int &pt, int &ot)
{
    // XCD x = bid & 7 takes runs of `chunk` consecutive pixel tiles (all their oc tiles), run r of the XCD
    // being global run 8*r + x.  chunk = 1 interleaves neighbouring tiles over the XCDs; a chunk of one or
    // more whole images keeps the lines an L2 has in flight contiguous in memory, which the write-bound
    // layers need (tools/probe_store_pattern2.hip: 4.2 -> 4.9 TB/s store-only at 28x28, 5.5 -> 5.8 at 56x56).
    const int bid = blockIdx.x;
    const int idx = bid >> 3;
    const int j = idx / a.n_oc_tiles;
    ot = idx - j * a.n_oc_tiles;
    const int c = j / a.chunk;
    pt = (c * 8 + (bid & 7)) * a.chunk + (j - c * a.chunk);
}

__device__ __forceinline__ bool decode_tile(const MfmaArgs &a, int &pt, int &ot, TileGeom &g, int &th)
{
    block_to_tile(a, pt, ot);
    if (pt >= a.n_pix_tiles) return false;
    const int ng = pt / a.tiles_h;
    g.n0 = ng * a.GI;
    g.oh0 = (pt - ng * a.tiles_h) * a.TH;
    th = min(a.TH, a.OH - g.oh0);
    g.OHWt = th * a.OW;
    g.NT = min(a.GI, a.N - g.n0) * g.OHWt;
    return true;
}

// ---------------------------------------------------------------------------------------------
// Main kernel.  WM x WN waves over (oc strips, pixel column tiles); NIW column tiles per wave;
// KKT = taps known at compile time (1, 9 = 3x3) or 0 for a runtime tap loop; X8 = 8-bit
// activations; NS = 32-channel chunks per stage: the 256 threads split into NS groups that each
// fetch one chunk of the stage, so one HBM round trip feeds NS*KK MFMA steps per column tile.
// ---------------------------------------------------------------------------------------------
template <int WM, int WN, int NIW, int KKT, bool X8, int NS, bool RQ = false, bool PATCH = false>
__global__ __launch_bounds__(MF_THREADS, 2) void conv_mfma_kernel(const MfmaArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint4 *Xs = reinterpret_cast<uint4 *>(smem);

    constexpr int MT = 32 * WM;
    constexpr int TPS = MF_THREADS / NS;   // staging threads per chunk
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int col = lane & 31, h = lane >> 5;
    const int sub = __builtin_amdgcn_readfirstlane(tid / TPS);   // which chunk of the stage this wave fetches

    int pt, ot, th;
    TileGeom g;
    if (!decode_tile(a, pt, ot, g, th)) return;
    const int NT = g.NT;
    const int ih0 = g.oh0 * a.stride - a.pad;
    const int ISZ = a.IHT * a.IWP;          // halo pixels of one image
    const int GSZ = a.GI * ISZ;             // halo pixels of one 16-channel group
    const int KK = (KKT > 0) ? KKT : a.KH * a.KW;
    const int trash = 2 * NS * GSZ + lane;
    int *sxp = reinterpret_cast<int *>(Xs + 2 * NS * GSZ + MF_TRASH);  // [GSZ] per-input-pixel sums (zw' != 0 only)

    // ---- zero the LDS halo image once: borders / padded channels are never written again ----
    for (int i = tid; i < 2 * NS * GSZ; i += MF_THREADS) Xs[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < GSZ; i += MF_THREADS) sxp[i] = 0;

    // ---- per-lane pixel bases of the wave's column tiles (uint4 index into Xs) ---------------
    const int RS = a.stride / a.ROWMUL, CS = a.stride / a.COLMUL;
    int pixidx[NIW];
#pragma unroll
    for (int t = 0; t < NIW; ++t) {
        const int q = (wn + t * WN) * 32 + col;
        const int gi = (a.GI > 1) ? q / g.OHWt : 0;
        const int rq = q - gi * g.OHWt;
        const int r = rq / a.OW, c = rq - r * a.OW;
        pixidx[t] = h * GSZ + ((q < NT) ? gi * ISZ + (r * RS) * a.IWP + c * CS : 0);
    }

    // ---- staging: thread <-> (image gi, halo row l, column quad iq) of chunk `sub`; its two units are
    // the two 16-channel groups of that chunk, so the channel of every load is wave-uniform (scalar
    // base + ONE per-thread 32-bit offset) and nothing needs a per-lane clamp. ------------------
    const int NQ = (a.W + 3) >> 2;
    const int HW = a.H * a.W;
    const int64_t img_off = (int64_t)g.n0 * a.IC * HW;                    // elements
    int u_off;                 // element offset from (image n0, channel 0) of (image gi, row ih, column iw0)
    int u_sh = 0;              // the row's last quad is read 4 bytes back from the row end and shifted
    int u_lds[4];              // uint4 index of pixel j in group 0 of the stage, or -1
    {
        const int lt = tid - sub * TPS;
        const int gi = lt / (a.IHT * NQ);
        const int rr = lt - gi * (a.IHT * NQ);
        const int l = rr / NQ, iq = rr - l * NQ;
        const int ih = ih0 + l * a.ROWMUL;
        const bool ok = gi < a.GI && g.n0 + gi < a.N && ih >= 0 && ih < a.H;
        int iw0 = 4 * iq;
        if (iw0 + 4 > a.W) { u_sh = 8 * (iw0 + 4 - a.W); iw0 = a.W - 4; }  // never read past the row (W >= 4)
        u_off = ok ? gi * a.IC * HW + ih * a.W + iw0 : 0;
        if (!ok) u_sh = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int iw = 4 * iq + j;
            const int cl = iw + a.pad;
            const int clc = cl / a.COLMUL;
            const bool pok = ok && iw < a.W && (clc * a.COLMUL == cl) && clc < a.IWP;
            u_lds[j] = pok ? (gi * a.IHT + l) * a.IWP + clc : -1;
        }
    }
    const uint8_t *xi = a.x + (X8 ? img_off : 0);   // 8-bit: image base (uniform); sub-8-bit: stream base
    const int64_t lim8 = a.x_bytes - 8;             // sub-8-bit: last byte offset of a full 8-byte read

    // ---- weight fragment pointer: lane (row col, half h) reads Wt[tap][2c + h][oc][16 B] ------
    const int8_t *a_base = a.wt + (int64_t)(ot * MT + wm * 32) * 16;  // wave-uniform
    const uint32_t a_voff = (uint32_t)(h * a.OCP + col) * 16u;           // per lane
    const int64_t grp_stride = (int64_t)a.OCP * 16;            // one 16-channel group
    const int64_t tap_stride = (int64_t)a.NG * grp_stride;     // one tap
    const int cbx = code_bias(a.x_bits, a.x_sign);

    v16i acc[NIW];
#pragma unroll
    for (int t = 0; t < NIW; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0;
    }

    // does any output channel of this tile have zw' != 0 ?  (workgroup-uniform)
    int zw_local = 0;
    if (tid < MT) zw_local = (a.ep[a.OCP + ot * MT + tid] != 0.0f) ? 1 : 0;
    const bool need_sx = __syncthreads_or(zw_local) != 0;  // also orders the LDS zero fill

#ifdef QE_STAMP
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = qe_stamp();
    const unsigned long long tstart = tprev;
#endif
    // Activation fetch.  No value is ever masked here: rows outside the image belong to threads
    // whose LDS writes go to the trash slots, and channels >= IC meet all-zero weights in Wt (prep).
    uint32_t d[MF_UNITS][16];
    auto issue_x = [&](int s) __attribute__((always_inline)) {
        const int c = s * NS + sub;
#pragma unroll
        for (int u = 0; u < MF_UNITS; ++u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int ic = c * 32 + u * 16 + i;                 // wave-uniform
                const int icc = ic < a.IC ? ic : a.IC - 1;
                if constexpr (X8) {
                    const uint8_t *plane = xi + (int64_t)icc * HW;  // scalar base
                    uint32_t v;
                    __builtin_memcpy(&v, plane + (uint32_t)u_off, 4);  // one (possibly unaligned) global_load_dword
                    d[u][i] = v;
                } else {
                    const int64_t bit = (img_off + (int64_t)icc * HW + u_off) * a.x_bits + (u_sh / 8) * a.x_bits;
                    const int64_t byte = bit >> 3;
                    const int64_t bc = byte < lim8 ? byte : lim8;
                    uint64_t v;
                    __builtin_memcpy(&v, xi + bc, 8);
                    v >>= ((int)(bit & 7) + 8 * (int)(byte - bc));
                    const uint32_t mask = (1u << a.x_bits) - 1u;
                    uint32_t r = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        r |= ((((uint32_t)(v >> (j * a.x_bits)) & mask) - (uint32_t)cbx) & 0xffu) << (8 * j);
                    d[u][i] = r;
                }
            }
        }
    };
    auto stage_x = [&](int s) __attribute__((always_inline)) {
        const int c = s * NS + sub;
#pragma unroll
        for (int u = 0; u < MF_UNITS; ++u) {
            if constexpr (X8) {
#pragma unroll
                for (int i = 0; i < 16; ++i) d[u][i] = (d[u][i] >> u_sh) ^ 0x80808080u;  // u - 128: signed q / unsigned q - 128
            }
            uint32_t o[4][4];
#pragma unroll
            for (int m = 0; m < 4; ++m)
                transpose4x4(d[u][4 * m], d[u][4 * m + 1], d[u][4 * m + 2], d[u][4 * m + 3],
                             o[0][m], o[1][m], o[2][m], o[3][m]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int idx = u_lds[j] >= 0 ? u_lds[j] + (sub * 2 + u) * GSZ : trash;
                Xs[idx] = make_uint4(o[j][0], o[j][1], o[j][2], o[j][3]);
            }
            if (need_sx) {
                // S_x needs sum_ic a_x per input pixel over REAL channels only
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int sum = 0;
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const int nv = a.IC - (c * 32 + u * 16 + 4 * m);  // valid channels in this dword
                        const int ones = nv >= 4 ? 0x01010101 : (nv <= 0 ? 0 : (0x01010101 & ((1 << (8 * nv)) - 1)));
                        sum = __builtin_amdgcn_sdot4((int)o[j][m], ones, sum, false);
                    }
                    if (u_lds[j] >= 0) atomicAdd(&sxp[u_lds[j]], sum);
                }
            }
        }
    };
    auto mma_tap = [&](const v4i af, int off) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < NIW; ++t) {
            const v4i b = *reinterpret_cast<const v4i *>(&Xs[pixidx[t] + off]);
            acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, b, acc[t], 0, 0, 0);
        }
    };
    // one stage of NS chunks: request A(s) -> X(s) regs into LDS -> barrier -> request X(s+1) -> MFMA
    auto stage = [&](int s, auto prefetch) __attribute__((always_inline)) {
        const int8_t *a_s = a_base + (int64_t)(2 * NS * s) * grp_stride;
        v4i afr[NS][(KKT > 0) ? KKT : 1];
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            if constexpr (KKT > 0) {
#pragma unroll
                for (int tap = 0; tap < KKT; ++tap)
                    afr[k][tap] = *reinterpret_cast<const v4i *>(a_s + 2 * k * grp_stride + tap * tap_stride + a_voff);
            } else {
                afr[k][0] = *reinterpret_cast<const v4i *>(a_s + 2 * k * grp_stride + a_voff);
            }
        }
        QE_ST(0);   // A requests issued
        stage_x(s);
        QE_ST(1);   // wait X + transpose + LDS writes
        __syncthreads();
        QE_ST(2);   // barrier 1
        if constexpr (decltype(prefetch)::value) issue_x(s + 1);
        QE_ST(3);   // X(s+1) issue
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            if constexpr (KKT > 0) {
#pragma unroll
                for (int tap = 0; tap < KKT; ++tap) {
                    constexpr int KW_T = (KKT == 9) ? 3 : 1;
                    mma_tap(afr[k][tap], 2 * k * GSZ + (tap / KW_T) * a.IWP + (tap % KW_T));
                }
            } else {
                v4i af = afr[k][0];
                for (int tap = 0; tap < KK; ++tap) {
                    const int nxt = tap + 1 < KK ? tap + 1 : tap;
                    const v4i af_next = *reinterpret_cast<const v4i *>(a_s + 2 * k * grp_stride + nxt * tap_stride + a_voff);
                    const int kh = tap / a.KW;
                    mma_tap(af, 2 * k * GSZ + kh * a.IWP + (tap - kh * a.KW));
                    af = af_next;
                }
            }
        }
        QE_ST(4);   // MFMA phase
        __syncthreads();  // everyone is done reading before the next stage overwrites the image
        QE_ST(5);   // barrier 2
    };

    const int n_stages = a.NCH / NS;   // NCH is padded to a multiple of NS by the host (zero weights)
    issue_x(0);
    for (int s = 0; s < n_stages - 1; ++s) stage(s, std::true_type{});
    stage(n_stages - 1, std::false_type{});

    // ---- epilogue -----------------------------------------------------------------------------
    int sxs[NIW];
#pragma unroll
    for (int t = 0; t < NIW; ++t) {
        sxs[t] = 0;
        if (need_sx) {
            const int pbase = pixidx[t] - h * GSZ;
            for (int tap = 0; tap < KK; ++tap) {
                const int kh = tap / a.KW;
                sxs[t] += sxp[pbase + kh * a.IWP + (tap - kh * a.KW)];  // zero outside the image
            }
        }
    }
#ifdef QE_STAMP
    QE_ST(6);       // (prologue of the epilogue)
#endif
    {
        const float *ctab = stage_ctab<32 * WM>(a, smem, ot, tid, MF_THREADS);
        const int *ptab = ctab ? nullptr : stage_ptab<32 * WM>(a, smem, ot, tid, MF_THREADS);
        if constexpr (RQ && PATCH) __syncthreads();   // every wave is done with the staging image and the pixel sums: the patch takes their place
        mfma_epilogue<WM, WN, NIW, RQ, PATCH>(a, acc, sxs, need_sx, g, ot, wm, wn, col, h, KK, ptab, ctab);
        if constexpr (RQ && PATCH) rq_patch_copy_out<32 * WM, 32 * NIW * WN, MF_THREADS>(a, g, ot, tid);
    }
#ifdef QE_STAMP
    QE_ST(7);       // epilogue stores issued
    if (a.dbg != nullptr && lane == 0) {
        unsigned long long *o = a.dbg + ((size_t)blockIdx.x * 4 + wave) * 10;
        for (int i = 0; i < 8; ++i) o[i] = st[i];
        o[8] = tprev - tstart;
        o[9] = tstart;
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// 3x3 kernel with TWO 32-row strips per wave and the weights in LDS ("sm2"), 8-bit activations.
// Why: in the halo kernel's 4x1 layout every wave reads all 7 activation fragments of a tap from
// LDS for 7 MFMAs: 4 waves x 7 KB per 224 MFMA cycles = 128 B/clk, the whole LDS bandwidth of the CU
// before a single bank conflict (and the pixel-major halo image has ~50 % 2-way conflicts at row
// breaks).  The matrix pipe therefore waited on ds_read (stamps: one MFMA per 58-68 cycles instead of 32).
// Here a wave owns 2 strips x 4 column tiles: one activation fragment feeds 2 MFMAs, so the same
// tile needs half the activation reads (4 waves x <= 4 KB per 256 MFMA cycles).  The price is that a
// strip's weight fragment is now wanted by WN waves; they come from an LDS copy of the chunk's
// weights (Wt order is already fragment order: straight 16-byte copies, conflict-free b128 reads)
// that is prefetched global -> VGPR during the previous chunk's MFMA phase exactly like the
// activations, so no wave ever waits for an L2 round trip inside the MFMA phase.
// Wave layout: WMS strip pairs x WN = 4 / WMS waves along the pixels (WMS = 2: 128 oc x <= 8 column
// tiles, WMS = 1: 64 oc x <= 16 column tiles); column tile of slot t = wn + t * WN.  A slot outside the
// tile (7 column tiles split 4 + 3) computes on pixel 0 and is never stored: a wave-uniform branch around
// it made hipcc spill 700-900 VGPRs (two live copies of the accumulators).
// Tile geometry, LDS halo image, staging threads and epilogue are the halo kernel's.
// ---------------------------------------------------------------------------------------------
template <int WMS, int KKT, int SPLIT, bool RQ = false, bool PATCH = false>
__global__ __launch_bounds__(MF_THREADS, 2) void conv_mfma_sm2_kernel(const MfmaArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint4 *Xs = reinterpret_cast<uint4 *>(smem);

    constexpr int MT = 64 * WMS;
    constexpr int WN = 4 / WMS;
    constexpr int NIW = 4;
    constexpr int KW_T = 3;
    constexpr int WPIECES = KKT * 2 * MT;                          // 16-byte weight pieces per chunk
    constexpr int PW = (WPIECES + MF_THREADS - 1) / MF_THREADS;    // per thread
    constexpr int SEG_PER_I = MF_THREADS / MT;                     // (tap, half) segments covered per piece round
    constexpr int TPS = MF_THREADS / SPLIT;                        // staging threads per channel slice
    constexpr int CPT = 32 / SPLIT;                                // channels a staging thread fetches per chunk
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wms = wave % WMS, wn = wave / WMS;
    const int col = lane & 31, h = lane >> 5;

    int pt, ot, th;
    TileGeom g;
    if (!decode_tile(a, pt, ot, g, th)) return;
    const int NT = g.NT;
    const int ih0 = g.oh0 * a.stride - a.pad;
    const int ISZ = a.IHT * a.IWP;
    const int GSZ = a.GI * ISZ;
    constexpr int KK = KKT;
    const int trash = 2 * GSZ + lane;
    int *sxp = reinterpret_cast<int *>(Xs + 2 * GSZ + MF_TRASH);
    uint4 *Ws = reinterpret_cast<uint4 *>(smem + ((((2 * GSZ + MF_TRASH) * 16 + GSZ * 4) + 15) & ~15));   // [KKT*2][MT] (+ slack to PW * 256 pieces)

    for (int i = tid; i < 2 * GSZ; i += MF_THREADS) Xs[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < GSZ; i += MF_THREADS) sxp[i] = 0;

    const int RS = a.stride, CS = a.stride;
    int pixidx[NIW];
#pragma unroll
    for (int t = 0; t < NIW; ++t) {
        const int q = (wn + t * WN) * 32 + col;
        const int gi = (a.GI > 1) ? q / g.OHWt : 0;
        const int rq = q - gi * g.OHWt;
        const int r = rq / a.OW, c = rq - r * a.OW;
        pixidx[t] = h * GSZ + ((q < NT) ? gi * ISZ + (r * RS) * a.IWP + c * CS : 0);
    }

    // ---- activation staging: thread <-> (channel slice, image gi, halo row l, column quad iq).  A tile has
    // only 60-90 (row, quad) units, so one wave used to do all the fetching, transposing and LDS writing of
    // a chunk while three waited at the barrier; SPLIT slices of 32 / SPLIT channels spread it over all four.
    const int NQ = (a.W + 3) >> 2;
    const int HW = a.H * a.W;
    const int64_t img_off = (int64_t)g.n0 * a.IC * HW;
    const int slice = __builtin_amdgcn_readfirstlane(tid / TPS);
    int u_off, u_sh = 0, u_lds[4];
    {
        const int lt = tid - slice * TPS;
        const int gi = lt / (a.IHT * NQ);
        const int rr = lt - gi * (a.IHT * NQ);
        const int l = rr / NQ, iq = rr - l * NQ;
        const int ih = ih0 + l;
        const bool ok = gi < a.GI && g.n0 + gi < a.N && ih >= 0 && ih < a.H;
        int iw0 = 4 * iq;
        if (iw0 + 4 > a.W) { u_sh = 8 * (iw0 + 4 - a.W); iw0 = a.W - 4; }
        u_off = ok ? gi * a.IC * HW + ih * a.W + iw0 : 0;
        if (!ok) u_sh = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int iw = 4 * iq + j;
            const int cl = iw + a.pad;
            const bool pok = ok && iw < a.W && cl < a.IWP;
            u_lds[j] = pok ? (gi * a.IHT + l) * a.IWP + cl : -1;
        }
    }
    const uint8_t *xi = a.x + img_off;

    // ---- weight staging: piece e = tid + 256 i <-> segment (tap, half) = e / MT, row e % MT of Wt ----
    const int seg0 = tid / MT, within = tid - seg0 * MT;
    const int tap0 = seg0 >> 1, hh0 = seg0 & 1;
    constexpr int TAP_PER_I = SEG_PER_I / 2;                       // taps advanced per piece round (1 or 2)
    const int64_t grp_stride = (int64_t)a.OCP * 16;
    const int64_t tap_stride = (int64_t)a.NG * grp_stride;
    const int8_t *w_thr = a.wt + ((int64_t)hh0 * a.OCP + ot * MT + within) * 16;   // + tap * tap_stride + 2s * grp_stride

    int zw_local = 0;
    if (tid < MT) zw_local = (a.ep[a.OCP + ot * MT + tid] != 0.0f) ? 1 : 0;
    const bool need_sx = __syncthreads_or(zw_local) != 0;

#ifdef QE_STAMP
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = qe_stamp();
    const unsigned long long tstart = tprev;
#endif
    uint32_t d[CPT];
    // weights of chunk s: LDS-DMA (global_load_lds_dwordx4), 16-byte pieces straight into Ws with no VGPR
    // round trip (a register-staged prefetch needs 36 more VGPRs than this kernel has: hipcc then keeps the
    // staging array in scratch and serialises the loads).  Lane l of a wave lands at base + 16 l, which is
    // exactly the piece order of Ws.  Rounds past the last tap (MT = 64) re-read tap 8 into the slack rows.
    auto issue_w = [&](int s) __attribute__((always_inline)) {
        const int8_t *ws = w_thr + (int64_t)(2 * s) * grp_stride;
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int tap = tap0 + i * TAP_PER_I;
            const int tapc = tap < KKT ? tap : KKT - 1;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(ws + (int64_t)tapc * tap_stride),
                (__attribute__((address_space(3))) void *)(Ws + MF_THREADS * i + 64 * wave), 16, 0, 0);
        }
    };
    auto issue_x = [&](int s) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) {
            const int ic = s * 32 + slice * CPT + i;               // wave-uniform
            const int icc = ic < a.IC ? ic : a.IC - 1;
            const uint8_t *plane = xi + (int64_t)icc * HW;
            uint32_t v;
            __builtin_memcpy(&v, plane + (uint32_t)u_off, 4);
            d[i] = v;
        }
    };
    auto stage_x = [&](int s) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < CPT; ++i) d[i] = (d[i] >> u_sh) ^ 0x80808080u;
        uint32_t o[4][CPT / 4];
#pragma unroll
        for (int m = 0; m < CPT / 4; ++m)
            transpose4x4(d[4 * m], d[4 * m + 1], d[4 * m + 2], d[4 * m + 3], o[0][m], o[1][m], o[2][m], o[3][m]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // pixel j: channels [slice * CPT, +CPT) of the chunk = bytes of the [pixel][16 ch] vector(s)
            if constexpr (CPT == 32) {
                const int idx = u_lds[j] >= 0 ? u_lds[j] : trash;
                Xs[idx] = make_uint4(o[j][0], o[j][1], o[j][2], o[j][3]);
                Xs[u_lds[j] >= 0 ? idx + GSZ : trash] = make_uint4(o[j][4], o[j][5], o[j][6], o[j][7]);
            } else if constexpr (CPT == 16) {
                const int idx = u_lds[j] >= 0 ? u_lds[j] + slice * GSZ : trash;
                Xs[idx] = make_uint4(o[j][0], o[j][1], o[j][2], o[j][3]);
            } else {
                const int idx = u_lds[j] >= 0 ? u_lds[j] + (slice >> 1) * GSZ : trash;
                *reinterpret_cast<uint2 *>(reinterpret_cast<uint8_t *>(&Xs[idx]) + (slice & 1) * 8) = make_uint2(o[j][0], o[j][1]);
            }
        }
        if (need_sx) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int sum = 0;
#pragma unroll
                for (int m = 0; m < CPT / 4; ++m) {
                    const int nv = a.IC - (s * 32 + slice * CPT + 4 * m);
                    const int ones = nv >= 4 ? 0x01010101 : (nv <= 0 ? 0 : (0x01010101 & ((1 << (8 * nv)) - 1)));
                    sum = __builtin_amdgcn_sdot4((int)o[j][m], ones, sum, false);
                }
                if (u_lds[j] >= 0) atomicAdd(&sxp[u_lds[j]], sum);
            }
        }
    };
    const uint4 *Wf = Ws + h * MT + (2 * wms) * 32 + col;    // + tap * 2 * MT (+ 32 for the second strip)
    // A slot outside the tile (7 column tiles split 4 + 3) computes on pixel 0 and is never stored.  Skipping it
    // was tried twice: a wave-uniform branch around it inside the chunk loop made hipcc keep two copies of the
    // accumulators (700-900 spilled VGPRs); two whole instances of the loop + epilogue selected once per wave
    // compiled cleanly but ran slower (56x56: 0.070 -> 0.091 ms, others unchanged).
    constexpr int NTC = NIW;
    v16i acc0[NIW], acc1[NIW];
#pragma unroll
    for (int t = 0; t < NIW; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[t][r] = 0; acc1[t][r] = 0; }
    }
    auto mma_chunk = [&]() __attribute__((always_inline)) {
        // fragments of tap k+1 are requested before the MFMAs of tap k (one tap of lookahead, 24 VGPRs)
        v4i af0, af1, b[NTC];
        auto fetch = [&](int tap, v4i &f0, v4i &f1, v4i (&bb)[NTC]) __attribute__((always_inline)) {
            const int off = (tap / KW_T) * a.IWP + (tap % KW_T);
            const uint4 w0 = Wf[tap * 2 * MT], w1 = Wf[tap * 2 * MT + 32];
            f0 = v4i{(int)w0.x, (int)w0.y, (int)w0.z, (int)w0.w};
            f1 = v4i{(int)w1.x, (int)w1.y, (int)w1.z, (int)w1.w};
#pragma unroll
            for (int t = 0; t < NTC; ++t) bb[t] = *reinterpret_cast<const v4i *>(&Xs[pixidx[t] + off]);
        };
        fetch(0, af0, af1, b);
#pragma unroll
        for (int tap = 0; tap < KKT; ++tap) {
            v4i nf0 = af0, nf1 = af1, nb[NTC];
#pragma unroll
            for (int t = 0; t < NTC; ++t) nb[t] = b[t];
            if (tap + 1 < KKT) fetch(tap + 1, nf0, nf1, nb);
#pragma unroll
            for (int t = 0; t < NTC; ++t) {
                acc0[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af0, b[t], acc0[t], 0, 0, 0);
                acc1[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af1, b[t], acc1[t], 0, 0, 0);
            }
            af0 = nf0; af1 = nf1;
#pragma unroll
            for (int t = 0; t < NTC; ++t) b[t] = nb[t];
            // issue order inside the tap: one fragment read of tap k+1 ahead of each of the first MFMAs of
            // tap k (left alone, hipcc sinks all six reads behind the seventh MFMA: 32 cycles of lookahead)
#pragma unroll
            for (int j = 0; j < NTC + 2; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * NTC - (NTC + 2), 0);
            __builtin_amdgcn_sched_barrier(0);   // keep later taps' fragment reads from piling up in registers
        }
    };
    // per chunk: X(s) registers -> LDS, W(s) landed -> barrier -> request X(s+1) -> MFMA -> barrier -> request W(s+1)
    // (single weight buffer: its refill has to wait for the last reader, and lands while the next stage_x runs)
    auto stage = [&](int s, auto prefetch) __attribute__((always_inline)) {
        stage_x(s);
        __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): the LDS-DMA of W(s) has landed
        QE_ST(1);   // wait X, W + transposes + LDS writes
        __syncthreads();
        QE_ST(2);   // barrier 1
        if constexpr (decltype(prefetch)::value) issue_x(s + 1);
        QE_ST(3);   // X(s+1) issue
        mma_chunk();
        QE_ST(4);   // MFMA phase
        __syncthreads();
        QE_ST(5);   // barrier 2
        if constexpr (decltype(prefetch)::value) issue_w(s + 1);
    };

    issue_x(0);
    issue_w(0);
    QE_ST(0);
    for (int s = 0; s < a.NCH - 1; ++s) stage(s, std::true_type{});
    stage(a.NCH - 1, std::false_type{});

    int sxs[NTC];
#pragma unroll
    for (int t = 0; t < NTC; ++t) {
        sxs[t] = 0;
        if (need_sx) {
            const int pbase = pixidx[t] - h * GSZ;
            for (int tap = 0; tap < KK; ++tap) {
                const int kh = tap / a.KW;
                sxs[t] += sxp[pbase + kh * a.IWP + (tap - kh * a.KW)];
            }
        }
    }
    QE_ST(6);
    const float *ctab = stage_ctab<MT>(a, smem, ot, tid, MF_THREADS);
    const int *ptab = ctab ? nullptr : stage_ptab<MT>(a, smem, ot, tid, MF_THREADS);
    if constexpr (RQ && PATCH) __syncthreads();
    mfma_epilogue<2 * WMS, WN, NTC, RQ, PATCH>(a, acc0, sxs, need_sx, g, ot, 2 * wms, wn, col, h, KK, ptab, ctab);
    mfma_epilogue<2 * WMS, WN, NTC, RQ, PATCH>(a, acc1, sxs, need_sx, g, ot, 2 * wms + 1, wn, col, h, KK, ptab, ctab);
    if constexpr (RQ && PATCH) rq_patch_copy_out<MT, 32 * NTC * WN, MF_THREADS>(a, g, ot, tid);
#ifdef QE_STAMP
    QE_ST(7);
    if (a.dbg != nullptr && lane == 0) {
        unsigned long long *o = a.dbg + ((size_t)blockIdx.x * 4 + wave) * 10;
        for (int i = 0; i < 8; ++i) o[i] = st[i];
        o[8] = tprev - tstart;
        o[9] = tstart;
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// Warp-specialised 3x3 kernel (8-bit activations): the MFMA-bound layers.
// In the halo kernel every wave alternates between fetch/transposes and MFMA, so a SIMD's matrix
// pipe idles during half of each wave's life (measured: MFMA phase 43 % of the wave time on
// 256->256 3x3 at 14x14).  Here a workgroup is 8 waves with fixed roles:
//   waves 0-3  consumers: weight fragments (L2 -> VGPR) + ds_read_b128 + v_mfma, nothing else.  Their
//              vmcnt queue only ever holds weight loads, so afr[tap] is re-requested for the NEXT
//              stage right after its last MFMA of this stage: the weights of a stage are always a
//              full stage old when they are needed, with no extra registers.
//   waves 4-7  producers: activation fetch (global -> VGPR), 4x4 byte transposes, LDS writes of stage
//              s+1 into the other half of a double-buffered halo image while stage s is consumed.
// One s_barrier per stage hands buffer (s+1)%2 to the consumers and buffer s%2 back to the producers.
// 512 threads x <= 256 VGPRs = one workgroup per CU, one consumer wave per SIMD (a single wave with 7
// independent accumulators can keep the matrix pipe issuing back to back).
// Tile, LDS image layout, operand roles and epilogue are the halo kernel's (4x1 consumer waves).
// ---------------------------------------------------------------------------------------------
template <int NIW, int KKT, int SPLIT, bool NOPAD, bool RQ = false>
__global__ __launch_bounds__(2 * MF_THREADS, 2) void conv_mfma_ws_kernel(const MfmaArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint4 *Xs = reinterpret_cast<uint4 *>(smem);

    constexpr int WM = 4, WN = 1, MT = 128;
    constexpr int KW_T = (KKT == 9) ? 3 : 1;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave < 4;
    const int cw = wave & 3;                 // index inside the role
    const int col = lane & 31, h = lane >> 5;

    int pt, ot, th;
    TileGeom g;
    if (!decode_tile(a, pt, ot, g, th)) return;
    const int NT = g.NT;
    const int ih0 = g.oh0 * a.stride - a.pad;
    // Unpadded-row mode (PADW = 0, stride-1 layers): LDS rows hold exactly W pixels, so the 32 lanes of a
    // column tile read 32 CONSECUTIVE 16-byte slots for every tap (conflict-free; with W+2-pixel rows the
    // 14- and 7-pixel-wide maps hit every bank twice and the kernel is LDS-bound).  A tap that would fall
    // left/right of the image then reads a neighbouring row's pixel: those lanes zero their fragment.
    const int GD = NOPAD ? a.pad : 0;        // guard slots in front of / behind each group image (PADW = 0 iff NOPAD)
    const int ISZ = a.IHT * a.IWP;
    const int GSZ = a.GI * ISZ + 2 * GD;     // slots of one 16-channel group
    const int BUF = 2 * GSZ;                 // uint4 slots of one buffer (2 groups = 32 channels)
    const int trash = 2 * BUF + lane;
    int *sxp = reinterpret_cast<int *>(Xs + 2 * BUF + MF_TRASH);

    for (int i = tid; i < 2 * BUF; i += 2 * MF_THREADS) Xs[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < GSZ; i += 2 * MF_THREADS) sxp[i] = 0;

    // asymmetric activations: the epilogue's border-class table is built now, while all eight waves are here
    const float *ctab = stage_ctab<MT>(a, smem, ot, tid, 2 * MF_THREADS);

    int zw_local = 0;
    if (tid < MT) zw_local = (a.ep[a.OCP + ot * MT + tid] != 0.0f) ? 1 : 0;
    const bool need_sx = __syncthreads_or(zw_local) != 0;  // also orders the LDS zero fill
    const int n_stages = a.NCH;
#ifdef QE_STAMP
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = qe_stamp();
    const unsigned long long tstart = tprev;
#endif

    if (!consumer) {
        // ================================ producers ==========================================
        // The 256 producer threads split a 32-channel chunk SPLIT ways (32 / SPLIT channels per thread):
        // small halo tiles (14x14: 64 pixel quads) would otherwise leave three of the four producer
        // waves idle and put all 32 loads + transposes of a stage on one wave.
        constexpr int CPT = 32 / SPLIT;            // channels per thread: 32, 16 or 8
        constexpr int TPS = MF_THREADS / SPLIT;    // threads per channel slice
        const int ptid = tid - MF_THREADS;
        const int sub = __builtin_amdgcn_readfirstlane(ptid / TPS);   // wave-uniform channel slice
        const int NQ = (a.W + 3) >> 2;
        const int HW = a.H * a.W;
        const uint8_t *xi = a.x + (int64_t)g.n0 * a.IC * HW;
        int u_off, u_sh = 0, u_lds[4];
        {
            const int lt = ptid - sub * TPS;
            const int gi = lt / (a.IHT * NQ);
            const int rr = lt - gi * (a.IHT * NQ);
            const int l = rr / NQ, iq = rr - l * NQ;
            const int ih = ih0 + l * a.ROWMUL;
            const bool ok = gi < a.GI && g.n0 + gi < a.N && ih >= 0 && ih < a.H;
            int iw0 = 4 * iq;
            if (iw0 + 4 > a.W) { u_sh = 8 * (iw0 + 4 - a.W); iw0 = a.W - 4; }
            u_off = ok ? gi * a.IC * HW + ih * a.W + iw0 : 0;
            if (!ok) u_sh = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int iw = 4 * iq + j;
                const int cl = iw + a.PADW;
                const int clc = cl / a.COLMUL;
                const bool pok = ok && iw < a.W && (clc * a.COLMUL == cl) && clc < a.IWP;
                u_lds[j] = pok ? GD + (gi * a.IHT + l) * a.IWP + clc : -1;
            }
        }
        const int ch0 = sub * CPT;                 // first channel of this thread's slice inside the chunk
        // Register ring: the fetch of stage s+1+D is requested when stage s+1 is written to LDS, so a
        // load has D stages (not one) to come back.  A 3x3 stage is ~1 us of MFMA work but an HBM/L2
        // round trip under load is 2-3 us: with a one-stage lead the whole workgroup ran at memory
        // latency per stage (measured).  Producers own few registers, so the ring is free.
        constexpr int D = (CPT == 32) ? 2 : 4;
        uint32_t d[D][CPT];
        auto issue_x = [&](int c, auto slot) __attribute__((always_inline)) {
            constexpr int S = decltype(slot)::value;
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                const int ic = c * 32 + ch0 + i;
                const int icc = ic < a.IC ? ic : a.IC - 1;
                uint32_t v;
                __builtin_memcpy(&v, xi + (int64_t)icc * HW + (uint32_t)u_off, 4);
                d[S][i] = v;
            }
        };
        auto stage_x = [&](int c, int buf, auto slot) __attribute__((always_inline)) {
            constexpr int S = decltype(slot)::value;
            uint32_t e[CPT];
#pragma unroll
            for (int i = 0; i < CPT; ++i) e[i] = (d[S][i] >> u_sh) ^ 0x80808080u;
            uint32_t o[4][CPT / 4];
#pragma unroll
            for (int m = 0; m < CPT / 4; ++m)
                transpose4x4(e[4 * m], e[4 * m + 1], e[4 * m + 2], e[4 * m + 3], o[0][m], o[1][m], o[2][m], o[3][m]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // byte address of this thread's channel slice inside the pixel's two 16-byte vectors
                if constexpr (CPT == 32) {
                    const int i0 = u_lds[j] >= 0 ? buf * BUF + u_lds[j] : trash;
                    const int i1 = u_lds[j] >= 0 ? buf * BUF + GSZ + u_lds[j] : trash;
                    Xs[i0] = make_uint4(o[j][0], o[j][1], o[j][2], o[j][3]);
                    Xs[i1] = make_uint4(o[j][4], o[j][5], o[j][6], o[j][7]);
                } else if constexpr (CPT == 16) {
                    const int i0 = u_lds[j] >= 0 ? buf * BUF + sub * GSZ + u_lds[j] : trash;
                    Xs[i0] = make_uint4(o[j][0], o[j][1], o[j][2], o[j][3]);
                } else {
                    const int i0 = u_lds[j] >= 0 ? buf * BUF + (sub >> 1) * GSZ + u_lds[j] : trash;
                    uint2 *p2 = reinterpret_cast<uint2 *>(&Xs[i0]) + (sub & 1);
                    *p2 = make_uint2(o[j][0], o[j][1]);
                }
            }
            if (need_sx) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int sum = 0;
#pragma unroll
                    for (int m = 0; m < CPT / 4; ++m) {
                        const int nv = a.IC - (c * 32 + ch0 + 4 * m);
                        const int ones = nv >= 4 ? 0x01010101 : (nv <= 0 ? 0 : (0x01010101 & ((1 << (8 * nv)) - 1)));
                        sum = __builtin_amdgcn_sdot4((int)o[j][m], ones, sum, false);
                    }
                    if (u_lds[j] >= 0) atomicAdd(&sxp[u_lds[j]], sum);
                }
            }
        };
        // Stage c always lives in ring slot c % D.  Chunk indices past the end are clamped by issue_x's
        // channel clamp (they re-read the last channels and are never staged), which keeps every load
        // of the steady-state loop unconditional.
        // prologue: stages 0..D-1 requested, stage 0 staged into buffer 0
        issue_x(0, std::integral_constant<int, 0>{});
        if constexpr (D > 1) issue_x(1, std::integral_constant<int, 1 % D>{});
        if constexpr (D > 2) issue_x(2, std::integral_constant<int, 2 % D>{});
        if constexpr (D > 3) issue_x(3, std::integral_constant<int, 3 % D>{});
        stage_x(0, 0, std::integral_constant<int, 0>{});
        issue_x(D, std::integral_constant<int, 0>{});
        QE_ST(0);                       // prologue
        __syncthreads();                // (P) buffer 0 is complete
        QE_ST(2);
        // consumer stage s <-> producer iteration s: write X(s+1) (slot (s+1)%D) into buffer (s+1)%2,
        // then request X(s+1+D) into the slot just freed.  Unrolled by D so the slots are static.
        auto iter = [&](int s, auto slot) __attribute__((always_inline)) {
            if (s + 1 < n_stages) stage_x(s + 1, (s + 1) & 1, slot);   // LDS work only behind the branch
            QE_ST(1);   // wait X + transpose + LDS writes
            issue_x(s + 1 + D, slot);
            QE_ST(3);   // X issue
            __syncthreads();
            QE_ST(2);   // barrier
        };
        int s = 0;
        for (; s + D <= n_stages; s += D) {
            iter(s, std::integral_constant<int, 1 % D>{});
            if constexpr (D > 1) iter(s + 1, std::integral_constant<int, 2 % D>{});
            if constexpr (D > 2) iter(s + 2, std::integral_constant<int, 3 % D>{});
            if constexpr (D > 3) iter(s + 3, std::integral_constant<int, 0>{});
        }
        // tail: n_stages % D remaining consumer stages
        if (s < n_stages) { iter(s, std::integral_constant<int, 1 % D>{}); ++s; }
        if constexpr (D > 2) {
            if (s < n_stages) { iter(s, std::integral_constant<int, 2 % D>{}); ++s; }
            if (s < n_stages) { iter(s, std::integral_constant<int, 3 % D>{}); ++s; }
        }
#ifdef QE_STAMP
        if (a.dbg != nullptr && lane == 0) {
            unsigned long long *o = a.dbg + ((size_t)blockIdx.x * 8 + wave) * 10;
            for (int i = 0; i < 8; ++i) o[i] = st[i];
            o[8] = tprev - tstart;
            o[9] = tstart;
        }
#endif
        return;
    }

    // ==================================== consumers ==========================================
    const int RS = a.stride / a.ROWMUL, CS = a.stride / a.COLMUL;
    int pixidx[NIW];
    unsigned km[NIW];        // bit kw set: tap column kw of this lane's pixel is inside the image row
#pragma unroll
    for (int t = 0; t < NIW; ++t) {
        const int q = t * 32 + col;
        const int gi = (a.GI > 1) ? q / g.OHWt : 0;
        const int rq = q - gi * g.OHWt;
        const int r = rq / a.OW, c = rq - r * a.OW;
        // origin of the receptive field: LDS column c*CS - GD (the guard slots absorb the -GD of row 0)
        pixidx[t] = h * GSZ + ((q < NT) ? gi * ISZ + (r * RS) * a.IWP + c * CS : 0);
        unsigned m = 0;
#pragma unroll
        for (int kw = 0; kw < KW_T; ++kw) {
            const int iw = c * a.stride - a.pad + kw;
            if (!NOPAD || (iw >= 0 && iw < a.W)) m |= 1u << kw;
        }
        km[t] = m;
    }
    const int8_t *a_base = a.wt + (int64_t)(ot * MT + cw * 32) * 16;   // wave-uniform
    const uint32_t a_voff = (uint32_t)(h * a.OCP + col) * 16u;
    const int64_t grp_stride = (int64_t)a.OCP * 16;
    const int64_t tap_stride = (int64_t)a.NG * grp_stride;

    v16i acc[NIW];
#pragma unroll
    for (int t = 0; t < NIW; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0;
    }
    v4i afr[KKT];
#pragma unroll
    for (int tap = 0; tap < KKT; ++tap) afr[tap] = *reinterpret_cast<const v4i *>(a_base + tap * tap_stride + a_voff);

    QE_ST(0);                           // consumer prologue (setup + first weight requests)
    __syncthreads();                    // (P) buffer 0 is complete
    QE_ST(2);
    // B fragments run one tap ahead of the MFMAs (7 ds_read_b128 interleaved 1:1 with the 7 MFMAs of
    // the previous tap): every LDS read is issued >= 7 MFMA slots before its use, so the single
    // consumer wave of a SIMD issues MFMAs back to back.
    v4i bq[2][NIW];
    for (int s = 0; s < n_stages; ++s) {
        const int boff = (s & 1) * BUF;
        const int sn = s + 1 < n_stages ? s + 1 : s;       // the last stage re-requests its own weights (9 L2 hits)
        const int8_t *a_n = a_base + (int64_t)(2 * sn) * grp_stride;
#pragma unroll
        for (int t = 0; t < NIW; ++t) bq[0][t] = *reinterpret_cast<const v4i *>(&Xs[pixidx[t] + boff]);
#pragma unroll
        for (int tap = 0; tap < KKT; ++tap) {
            const int cur = tap & 1, nxt = cur ^ 1;
            if (tap + 1 < KKT) {
                const int off = boff + ((tap + 1) / KW_T) * a.IWP + ((tap + 1) % KW_T);
#pragma unroll
                for (int t = 0; t < NIW; ++t) bq[nxt][t] = *reinterpret_cast<const v4i *>(&Xs[pixidx[t] + off]);
            }
#pragma unroll
            for (int t = 0; t < NIW; ++t) {
                v4i bf = bq[cur][t];
                if constexpr (NOPAD && (KW_T == 3)) {
                    if ((tap % KW_T) != 1) {             // 3x3 / pad 1: the centre column is always inside
                        const int keep = ((km[t] >> (tap % KW_T)) & 1u) ? -1 : 0;
#pragma unroll
                        for (int k = 0; k < 4; ++k) bf[k] &= keep;
                    }
                }
                acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(afr[tap], bf, acc[t], 0, 0, 0);
            }
            afr[tap] = *reinterpret_cast<const v4i *>(a_n + tap * tap_stride + a_voff);   // next stage's fragment
            if (tap + 1 < KKT) {
#pragma unroll
                for (int t = 0; t < NIW; ++t) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // taps stay in order: the scheduler may not regroup the 63 MFMAs
        }
        QE_ST(4);   // MFMA phase
        __syncthreads();
        QE_ST(5);   // barrier
    }

    int sxs[NIW];
#pragma unroll
    for (int t = 0; t < NIW; ++t) {
        sxs[t] = 0;
        if (need_sx) {
            const int pbase = pixidx[t] - h * GSZ;
            for (int tap = 0; tap < KKT; ++tap)
                if ((km[t] >> (tap % KW_T)) & 1u) sxs[t] += sxp[pbase + (tap / KW_T) * a.IWP + (tap % KW_T)];
        }
    }
    mfma_epilogue<WM, WN, NIW, RQ>(a, acc, sxs, need_sx, g, ot, cw, 0, col, h, KKT, nullptr, ctab);
#ifdef QE_STAMP
    QE_ST(7);
    if (a.dbg != nullptr && lane == 0) {
        unsigned long long *o = a.dbg + ((size_t)blockIdx.x * 8 + wave) * 10;
        for (int i = 0; i < 8; ++i) o[i] = st[i];
        o[8] = tprev - tstart;
        o[9] = tstart;
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// Small-IC variant (IC <= 4, KW <= 8, 8-bit activations): the stem convolution (3 -> 64, 7x7/2).
// Padding 3 channels to a 32-channel chunk would waste 10x the MFMA work and LDS, so K is laid out
// as (kh) x [kw 0..7][ic 0..3]: one 32-deep MFMA step per kernel row.  LDS holds the halo tile
// as one dword per pixel [c0 c1 c2 c3]; the B fragment of lane (pixel, half h) is the 16 bytes of
// 4 consecutive pixels starting at column ow*stride + 4h of row oh*stride + kh.
//   Wt layout here: [KH][2][OCP][16], byte (kw - 4h)*4 + ic.
// ---------------------------------------------------------------------------------------------
template <int WM, int WN, int NIW, bool RQ = false, bool PATCH = false>
__global__ __launch_bounds__(MF_THREADS, 2) void conv_mfma_smallic_kernel(const MfmaArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *Xs = reinterpret_cast<uint32_t *>(smem);

    constexpr int MT = 32 * WM;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int col = lane & 31, h = lane >> 5;

    int pt, ot, th;
    TileGeom g;
    if (!decode_tile(a, pt, ot, g, th)) return;
    const int n = g.n0, oh0 = g.oh0, NT = g.NT;
    const int ih0 = oh0 * a.stride - a.pad;
    const int GSZ = a.IHT * a.IWP;       // pixels (dwords) in the halo image
    const int KK = a.KH * a.KW;
    int *sxp = reinterpret_cast<int *>(Xs + GSZ + MF_TRASH);
    const int trash = GSZ + lane;

    for (int i = tid; i < GSZ; i += MF_THREADS) { Xs[i] = 0; sxp[i] = 0; }

    int zw_local = 0;
    if (tid < MT) zw_local = (a.ep[a.OCP + ot * MT + tid] != 0.0f) ? 1 : 0;
    const bool need_sx = __syncthreads_or(zw_local) != 0;

    // weight fragments of all kernel rows (L2 hits), requested before the activation loads
    v4i afr[8];
    {
        const int8_t *a_base = a.wt + (int64_t)(ot * MT + wm * 32) * 16;
        const uint32_t a_voff = (uint32_t)(h * a.OCP + col) * 16u;
#pragma unroll
        for (int kh = 0; kh < 8; ++kh) {
            const int khc = kh < a.KH ? kh : a.KH - 1;
            afr[kh] = *reinterpret_cast<const v4i *>(a_base + (int64_t)khc * 2 * a.OCP * 16 + a_voff);
        }
    }

    // ---- stage the halo tile: thread <-> (row l, quad iq), IC dwords -> 4 pixel dwords ----------
    const int NQ = (a.W + 3) >> 2;
    const int HW = a.H * a.W;
    const uint8_t *xi = a.x + (int64_t)n * a.IC * HW;
    const int ones = a.IC >= 4 ? 0x01010101 : (0x01010101 & ((1 << (8 * a.IC)) - 1));
    // All loads of up to 4 units per thread are issued before the first one is consumed (a plain loop over the
    // units waits for each unit's loads before it requests the next: three dependent HBM round trips per tile).
    const int n_units = a.IHT * NQ;
    for (int u0 = 0; u0 < n_units; u0 += 4 * MF_THREADS) {
        uint32_t dch[4][4];
        int ul[4], uq[4], ush[4];
        bool uok[4], ulive[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int uid = u0 + tid + it * MF_THREADS;
            ulive[it] = uid < n_units;
            const int uc = ulive[it] ? uid : 0;
            const int l = uc / NQ, iq = uc - l * NQ;
            const int ih = ih0 + l;
            const bool ok = ih >= 0 && ih < a.H;
            int iw0 = 4 * iq, sh = 0;
            if (iw0 + 4 > a.W) { sh = 8 * (iw0 + 4 - a.W); iw0 = a.W - 4; }
            const uint32_t off = ok ? (uint32_t)(ih * a.W + iw0) : 0u;
            ul[it] = l; uq[it] = iq; ush[it] = sh; uok[it] = ok;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int icc = i < a.IC ? i : a.IC - 1;       // uniform
                uint32_t v;
                __builtin_memcpy(&v, xi + (int64_t)icc * HW + off, 4);
                dch[it][i] = v;
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            uint32_t c4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) c4[i] = (i < a.IC) ? ((dch[it][i] >> ush[it]) ^ 0x80808080u) : 0u;  // padded channel = 0
            uint32_t o0, o1, o2, o3;
            transpose4x4(c4[0], c4[1], c4[2], c4[3], o0, o1, o2, o3);
            const uint32_t o[4] = {o0, o1, o2, o3};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int iw = 4 * uq[it] + j;
                const int cl = iw + a.pad;
                const bool pok = ulive[it] && uok[it] && iw < a.W && cl < a.IWP;
                const int idx = pok ? ul[it] * a.IWP + cl : trash;
                Xs[idx] = o[j];
                if (need_sx && pok) sxp[idx] = __builtin_amdgcn_sdot4((int)o[j], ones, 0, false);
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

    v16i acc[NIW];
#pragma unroll
    for (int t = 0; t < NIW; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0;
    }
#pragma unroll
    for (int kh = 0; kh < 8; ++kh) {
        if (kh < a.KH) {
#pragma unroll
            for (int t = 0; t < NIW; ++t) {
                const uint32_t *bp = &Xs[pixidx[t] + kh * a.IWP + 4 * h];
                v4i b;
                b[0] = (int)bp[0]; b[1] = (int)bp[1]; b[2] = (int)bp[2]; b[3] = (int)bp[3];
                acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(afr[kh], b, acc[t], 0, 0, 0);
            }
        }
    }

    int sxs[NIW];
#pragma unroll
    for (int t = 0; t < NIW; ++t) {
        sxs[t] = 0;
        if (need_sx) {
            for (int tap = 0; tap < KK; ++tap) {
                const int kh = tap / a.KW;
                sxs[t] += sxp[pixidx[t] + kh * a.IWP + (tap - kh * a.KW)];
            }
        }
    }
    {
        const float *ctab = stage_ctab<32 * WM>(a, reinterpret_cast<uint8_t *>(smem), ot, tid, MF_THREADS);
        const int *ptab = ctab ? nullptr : stage_ptab<32 * WM>(a, reinterpret_cast<uint8_t *>(smem), ot, tid, MF_THREADS);
        if constexpr (RQ && PATCH) __syncthreads();
        mfma_epilogue<WM, WN, NIW, RQ, PATCH>(a, acc, sxs, need_sx, g, ot, wm, wn, col, h, KK, ptab, ctab);
        if constexpr (RQ && PATCH) rq_patch_copy_out<32 * WM, 32 * NIW * WN, MF_THREADS>(a, g, ot, tid);
    }
}


// ---------------------------------------------------------------------------------------------
// Flat 1x1 kernel: KH = KW = 1, stride 1, padding 0, 8-bit activations, H*W % 4 == 0.
// A 1x1 convolution is a plain GEMM per image: out[oc][p] = sum_c W[oc][c] X[c][p] over the FLAT
// pixel index p of the NCHW planes.  Nothing has to be transposed by hand here:
//   * activations go global -> VGPR -> LDS in their native [channel][pixel] order as 16-byte pieces
//     (16 pixels of one channel per lane: 4.5x fewer load instructions than the halo kernel's
//     dword fetch, no v_perm work);
//   * MFMA operands want 16 K-contiguous bytes per lane: ds_read_b64_tr_b8 (gfx950's transposed LDS
//     read, semantics probed in tools/probe_tr_b8.hip: lane i of a 16-lane group receives column i
//     of an 8-row x 16-byte block whose row r address comes from lanes 2r, 2r+1) delivers exactly
//     that from the row-major image, two reads per fragment;
//   * the row stride is 32*(odd) bytes, so the 8 rows of a transposed read hit 8 distinct 32-byte
//     bank groups: conflict free;
//   * operand roles are swapped w.r.t. the halo kernel: A = activations (rows = pixels), B = weights
//     (cols = output channel).  D then has the output channel on the lane and 4 consecutive pixels
//     in 4 consecutive registers: every store is a 16-byte global_store_dwordx4 (4x fewer store
//     instructions), and alpha / bias / zero-point terms are ONE value per lane;
//   * 8-bit weights with IC % 16 == 0 are consumed straight from the packed OIHW tensor (row oc,
//     16 channels = 16 contiguous bytes, u ^ 0x80 in registers): no prep launch for these layers.
//     S_w (sum of weights per output channel, needed when zx' != 0) falls out of the same
//     fragments with v_dot4.
// Template: WM x WN waves (oc strips x pixel tiles), NIW pixel tiles per wave, NS 32-channel chunks
// per stage, WRAW = weights read from the packed tensor (else from the prep'd Wt).
// ---------------------------------------------------------------------------------------------
typedef int v2i __attribute__((ext_vector_type(2)));

// X4: 4-bit activations consumed straight from the packed stream (a piece = 16 pixels = 8 bytes, nibbles spread to bytes
// in the staging registers) instead of being expanded to 8-bit codes by a pass of their own first.
// NSTAGES > 0 (8-wave instances): the K loop fully unrolled over exactly NSTAGES stages with the activations requested TWO
// stages ahead (two register sets of PPT pieces: 4 each with 512 threads) and the weights one stage ahead, so a request is in
// flight at every moment of a stage -- with one set the time from a stage's arrival to the next request (LDS writes, barrier)
// plus a full memory round trip is serial in every stage.
template <int WM, int WN, int NIW, int NS, bool WRAW, bool S2, bool X4 = false, int NSTAGES = 0>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN > 4) ? 1 : 2) void conv_mfma_flat_kernel(const MfmaArgs a)
{
    constexpr int THR = 64 * WM * WN;           // 4 waves; 8 (WM = 8: 256 output channels per workgroup, one workgroup per CU)
    static_assert(!(X4 && S2), "the stride-2 staging takes 8-bit codes");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    constexpr int MT = 32 * WM;
    constexpr int NTILES = NIW * WN;
    constexpr int NTP = 32 * NTILES;            // pixels per tile
    constexpr int RSTR = 32 * (NTILES | 1);     // LDS row stride: odd multiple of 32 B
    constexpr int CK = 32 * NS;                 // channels per stage
    constexpr int SEGS = NTP / 16;              // 16-pixel pieces per channel row
    // S2 (1x1, stride 2, no padding): the GEMM runs over the flat OUTPUT pixels; a piece is 16 input bytes of
    // an even input row, of which the 8 even columns are kept.  Pieces per thread then depend on the row
    // geometry (rows x segments of the tile): 8 slots cover the supported shapes (host checks).
    constexpr int PPT = S2 ? 8 : (CK * SEGS + THR - 1) / THR;  // pieces per thread per stage

#ifdef QE_STAMP
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = qe_stamp();
    const unsigned long long tstart = tprev;
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int col = lane & 31, h = lane >> 5;

    const int P = S2 ? a.OH * a.OW : a.H * a.W;   // output pixels per image (== input pixels unless S2)
    const int PIN = a.H * a.W;                    // input plane
    int pt, ot;
    block_to_tile(a, pt, ot);
    if (pt >= a.n_pix_tiles) return;
    const int n = pt / a.tiles_h;               // tiles_h = pixel tiles per image here
    const int p0 = (pt - n * a.tiles_h) * NTP;
    const int NT = min(NTP, P - p0);            // valid pixels of this tile (multiple of 4)

    uint8_t *Xs = smem;                                                   // [CK][RSTR]
    constexpr int XS_BYTES = (CK * RSTR > WM * WN * 32 * 36 * 4) ? CK * RSTR : WM * WN * 32 * 36 * 4;  // image, later the epilogue patches
    int *sxp = reinterpret_cast<int *>(smem + XS_BYTES);                  // [NTP], only when zw' != 0

    // ---- epilogue constants of this lane's output channel -----------------------------------
    const int oc = ot * MT + wm * 32 + col;
    const int occ = oc < a.OC ? oc : a.OC - 1;
    const float sw = a.w_per_tensor ? a.w_scale[0] : a.w_scale[occ];
    const float zwp = (a.w_per_tensor ? a.w_zero[0] : a.w_zero[occ]) - zero_shift(a.w_bits, a.w_sign);
    const float alpha = a.x_scale[0] * sw;
    const float bia = a.bias ? a.bias[occ] : 0.0f;
    const float zxp = a.x_zero[0] - zero_shift(a.x_bits, a.x_sign);

    // ---- staging pieces of this thread (stage invariant) -------------------------------------
    // piece e = tid + 256*i <-> (channel c = e / SEGS of the stage, 16-pixel segment s = e % SEGS).
    // The read is clamped to end at the plane's end (P >= 16); a clamped piece is rotated back by
    // whole dwords (P % 4 == 0), so no read ever leaves the tensor.
    const int64_t img = (int64_t)n * a.IC * PIN;
    const uint8_t *xi = a.x + (X4 ? img / 2 : img);     // X4: two pixels per byte (PIN % 4 == 0: images start on a byte)
    int pc[PPT];          // channel within the stage
    int poff[PPT];        // clamped pixel offset inside the plane
    int prot[PPT];        // dwords to rotate (0 = piece was not clamped)
    int plds[PPT];        // LDS byte offset, or -1 when the piece does not exist
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int e = tid + THR * i;
        if constexpr (!S2) {
            const int c = e / SEGS, sg = e - c * SEGS;
            const int px = p0 + 16 * sg;
            const bool ok = c < CK && 16 * sg < NT;
            const int pxc = px < P - 16 ? px : P - 16;
            pc[i] = c < CK ? c : 0;
            poff[i] = ok ? pxc : 0;
            prot[i] = ok ? (px - pxc) >> 2 : 0;
            plds[i] = ok ? c * RSTR + 16 * sg : -1;
        } else {
            // e <-> (channel c, output row r of the tile, 16-column input segment sg)
            const int RT = NTP / a.OW;                       // output rows per tile (NTP % OW == 0)
            const int SEG = (a.W + 15) >> 4;                 // 16-byte segments per input row
            const int c = e / (RT * SEG);
            const int rem2 = e - c * (RT * SEG);
            const int r = rem2 / SEG, sg = rem2 - r * SEG;
            const int oh = p0 / a.OW + r;
            const bool ok = c < CK && oh < a.OH;
            const int iw = 16 * sg;
            const int iwc = iw < a.W - 16 ? iw : a.W - 16;   // never read past the row (W >= 16, W % 4 == 0)
            pc[i] = c < CK ? c : 0;
            poff[i] = ok ? (2 * oh) * a.W + iwc : 0;
            prot[i] = ok ? (iw - iwc) >> 2 : 0;
            // 8 output pixels of the piece start at column 8*sg; OW % 4 == 0 keeps both dwords whole
            plds[i] = ok ? c * RSTR + r * a.OW + 8 * sg : -1;
        }
    }

    // ---- weight fragment addressing -----------------------------------------------------------
    // WRAW: lane (oc, half h) reads 16 contiguous bytes of row oc of the packed OIHW tensor.
    // else: prep'd Wt[0][icg][OCP][16].
    const int NGR = (a.IC + 15) >> 4;           // real 16-channel groups
    const uint8_t *w_lane = WRAW ? a.w_raw + (int64_t)occ * a.IC : nullptr;
    const int8_t *wt_base = a.wt + (int64_t)(ot * MT + wm * 32) * 16;
    const uint32_t wt_voff = (uint32_t)col * 16u;

    v16i acc[NIW];
#pragma unroll
    for (int t = 0; t < NIW; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0;
    }
    int swacc = 0;        // sum of this lane's weight bytes (its half of every chunk)

    // transposed-read base of this lane: row (i>>1) of an 8-row block, column half (i&1), 16-pixel
    // group (lane>>4)&1 of the 32-pixel tile, channel half h of the 32-channel chunk
    const int i16 = lane & 15;
    const int tr_base = (16 * h + (i16 >> 1)) * RSTR + 16 * ((lane >> 4) & 1) + 8 * (i16 & 1);

    using DT = std::conditional_t<X4, uint2, uint4>;
    DT d[PPT];
    auto issue_x = [&](int s, DT (&d)[PPT]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int cg = s * CK + pc[i];
            const int cgc = cg < a.IC ? cg : a.IC - 1;
            if constexpr (X4) {
                const uint8_t *src = xi + ((uint32_t)(cgc * PIN + poff[i]) >> 1);   // element offsets are multiples of 4
                __builtin_memcpy(&d[i], src, 8);  // one (2-byte aligned) global_load_dwordx2 = 16 pixels
            } else {
                const uint8_t *src = xi + (uint32_t)(cgc * PIN + poff[i]);
                __builtin_memcpy(&d[i], src, 16);     // one (4-byte aligned) global_load_dwordx4
            }
        }
    };
    issue_x(0, d);   // in flight while the zero-point test below synchronises the workgroup

    const bool need_sx = __syncthreads_or((oc < a.OC && zwp != 0.0f) ? 1 : 0) != 0;   // workgroup-uniform
    if (need_sx) {
        for (int i = tid; i < NTP; i += THR) sxp[i] = 0;
        __syncthreads();
    }

    auto stage_x = [&](int s, DT (&d)[PPT]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            uint32_t v0, v1, v2, v3;
            if constexpr (X4) {
                // 16 nibbles -> 16 bytes (4 pixels per dword), then stored code u -> operand a = u - 8 (signed) | u
                auto spread = [](uint32_t h16) __attribute__((always_inline)) -> uint32_t {
                    uint32_t x = h16 & 0xffffu;
                    x = (x | (x << 8)) & 0x00ff00ffu;
                    return (x | (x << 4)) & 0x0f0f0f0fu;
                };
                auto to_op = [&](uint32_t u) __attribute__((always_inline)) -> uint32_t {
                    if (!a.x_sign) return u;
                    const uint32_t t = u ^ 0x08080808u;                     // q >= 0: t = q; q < 0: t = q + 16
                    return t | ((t & 0x08080808u) * 30u);                   // ... sign-extend the nibble (8 * 30 = 0xF0 per byte)
                };
                v0 = to_op(spread(d[i].x)); v1 = to_op(spread(d[i].x >> 16));
                v2 = to_op(spread(d[i].y)); v3 = to_op(spread(d[i].y >> 16));
            } else {
                v0 = d[i].x; v1 = d[i].y; v2 = d[i].z; v3 = d[i].w;
            }
            // rotate a clamped tail piece back: out[j] = in[(j + rot) & 3]; rot is 0 almost everywhere
            const int rot = prot[i];
            const uint32_t r0 = rot == 0 ? v0 : (rot == 1 ? v1 : (rot == 2 ? v2 : v3));
            const uint32_t r1 = rot == 0 ? v1 : (rot == 1 ? v2 : (rot == 2 ? v3 : v0));
            const uint32_t r2 = rot == 0 ? v2 : (rot == 1 ? v3 : (rot == 2 ? v0 : v1));
            const uint32_t r3 = rot == 0 ? v3 : (rot == 1 ? v0 : (rot == 2 ? v1 : v2));
            constexpr uint32_t XR = X4 ? 0u : 0x80808080u;      // 8-bit codes: u - 128 (signed q, or unsigned q - 128)
            const uint4 w4 = make_uint4(r0 ^ XR, r1 ^ XR, r2 ^ XR, r3 ^ XR);
            if constexpr (S2) {
                if (plds[i] >= 0) {
                    // keep the even columns: bytes 0,2 of each dword
                    const uint32_t e0 = __builtin_amdgcn_perm(w4.y, w4.x, 0x06040200u);
                    const uint32_t e1 = __builtin_amdgcn_perm(w4.w, w4.z, 0x06040200u);
                    const int colb = plds[i] - pc[i] * RSTR;             // pixel offset inside the tile
                    const int ow0 = colb % a.OW;
                    if ((a.OW & 3) == 0) {
                        uint32_t *dst = reinterpret_cast<uint32_t *>(Xs + plds[i]);
                        if (ow0 < a.OW) dst[0] = e0;
                        if (ow0 + 4 < a.OW) dst[1] = e1;
                    } else {
                        // rows of the LDS image are not dword aligned (OW = 14): byte stores
                        for (int j = 0; j < 8; ++j)
                            if (ow0 + j < a.OW) Xs[plds[i] + j] = (uint8_t)((j < 4 ? e0 : e1) >> (8 * (j & 3)));
                    }
                    if (need_sx && s * CK + pc[i] < a.IC) {
                        for (int j = 0; j < 8; ++j)
                            if (ow0 + j < a.OW)
                                atomicAdd(&sxp[colb + j], (int)(int8_t)((j < 4 ? e0 : e1) >> (8 * (j & 3))));
                    }
                }
            } else
            if (plds[i] >= 0) {
                *reinterpret_cast<uint4 *>(Xs + plds[i]) = w4;
                if (need_sx && s * CK + pc[i] < a.IC) {
                    // rare path (asymmetric weights): S_x[p] = sum over real channels of a_x
                    const int pb = (plds[i] - pc[i] * RSTR);
                    const uint32_t ww[4] = {w4.x, w4.y, w4.z, w4.w};
                    for (int j = 0; j < 16; ++j)
                        atomicAdd(&sxp[pb + j], (int)(int8_t)(ww[j >> 2] >> (8 * (j & 3))));
                }
            }
        }
    };
    auto stage = [&](int s, auto prefetch) __attribute__((always_inline)) {
        // (1) weight fragments of the stage's NS chunks
        v4i wf[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int icg = 2 * (s * NS + k) + h;                 // 16-channel group of this lane half
            const int icgc = icg < NGR ? icg : NGR - 1;
            v4i f;
            if constexpr (WRAW) {
                __builtin_memcpy(&f, w_lane + icgc * 16, 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) f[j] ^= (int)0x80808080;   // u - 128: signed q, or unsigned q - 128
            } else {
                f = *reinterpret_cast<const v4i *>(wt_base + (int64_t)icgc * a.OCP * 16 + wt_voff);
            }
            if (icg >= NGR) f = v4i{0, 0, 0, 0};                  // channel padding of the last chunk
            wf[k] = f;
        }
        QE_ST(0);   // prologue / weight requests
        // (2) activations of this stage: registers -> LDS
        stage_x(s, d);
        QE_ST(1);   // wait X + LDS writes
        __syncthreads();
        QE_ST(2);   // barrier 1
        // (3) next stage's activations in flight under the MFMA phase
        if constexpr (decltype(prefetch)::value) issue_x(s + 1, d);
        QE_ST(3);   // X(s+1) issue
        // (4) MFMA: A = transposed activation fragment (rows = pixels), B = weights (cols = oc)
#pragma unroll
        for (int k = 0; k < NS; ++k) {
#pragma unroll
            for (int j = 0; j < 4; ++j) swacc = __builtin_amdgcn_sdot4(wf[k][j], 0x01010101, swacc, false);
#pragma unroll
            for (int t = 0; t < NIW; ++t) {
                const uint8_t *src = Xs + tr_base + (k * 32) * RSTR + (wn + t * WN) * 32;
                const v2i lo = __builtin_amdgcn_ds_read_tr8_b64_v2i32(
                    (v2i __attribute__((address_space(3))) *)(src));
                const v2i hi = __builtin_amdgcn_ds_read_tr8_b64_v2i32(
                    (v2i __attribute__((address_space(3))) *)(src + 8 * RSTR));
                const v4i xf = {lo[0], lo[1], hi[0], hi[1]};
                acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xf, wf[k], acc[t], 0, 0, 0);
            }
        }
        QE_ST(4);   // MFMA phase (incl. weight wait)
        __syncthreads();
        QE_ST(5);   // barrier 2
    };

    if constexpr (NSTAGES > 0) {
        static_assert(WRAW && !S2 && !X4, "deep-prefetch form: packed 8-bit weights, stride 1");
        DT d2[PPT];
        v4i wfa[NS], wfb[NS];
        auto load_w = [&](int s, v4i (&wf)[NS]) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < NS; ++k) __builtin_memcpy(&wf[k], w_lane + (2 * (s * NS + k) + h) * 16, 16);   // IC == NSTAGES * CK: no padding
        };
        auto mma = [&](v4i (&wf)[NS]) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                v4i f = wf[k];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f[j] ^= (int)0x80808080;
                    swacc = __builtin_amdgcn_sdot4(f[j], 0x01010101, swacc, false);
                }
#pragma unroll
                for (int t = 0; t < NIW; ++t) {
                    const uint8_t *src = Xs + tr_base + (k * 32) * RSTR + (wn + t * WN) * 32;
                    const v2i lo = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3))) *)(src));
                    const v2i hi = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3))) *)(src + 8 * RSTR));
                    const v4i xf = {lo[0], lo[1], hi[0], hi[1]};
                    acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xf, f, acc[t], 0, 0, 0);
                }
            }
        };
        if constexpr (NSTAGES > 1) issue_x(1, d2);
        load_w(0, wfa);
        QE_ST(0);
#pragma unroll
        for (int s = 0; s < NSTAGES; ++s) {
            if (s & 1) stage_x(s, d2); else stage_x(s, d);
            QE_ST(1);
            __syncthreads();
            QE_ST(2);
            if (s + 2 < NSTAGES) { if (s & 1) issue_x(s + 2, d2); else issue_x(s + 2, d); }
            if (s + 1 < NSTAGES) { if (s & 1) load_w(s + 1, wfa); else load_w(s + 1, wfb); }
            QE_ST(3);
            if (s & 1) mma(wfb); else mma(wfa);
            QE_ST(4);
            __syncthreads();
            QE_ST(5);
        }
    } else {
    const int n_stages = (a.IC + CK - 1) / CK;
    for (int s = 0; s < n_stages - 1; ++s) stage(s, std::true_type{});
    stage(n_stages - 1, std::false_type{});
    }

    // ---- epilogue: lane = output channel, 4 consecutive registers = 4 consecutive pixels ------
    const int sw_sum = swacc + __shfl_xor(swacc, 32);   // both channel halves
    // out = bias + alpha * (S_aw - zw' S_x - zx' S_w + IC zx' zw')   (1x1, no padding: every tap in bounds)
    const float cst = fmaf((float)a.IC * zxp, zwp, -zxp * (float)sw_sum);
    // D holds 4 consecutive pixels of ONE output channel per lane; stored as-is every instruction
    // would touch 32 channel rows x 32 bytes (partial 128-byte lines: measured slower than dword
    // stores of full lines).  Each wave therefore turns its 32 oc x 32 px tile through a private LDS
    // patch (row stride 36 floats: conflict-free b128 writes) and reads it back with 8 lanes per
    // channel row: one global_store_dwordx4 then writes 8 rows x 128 contiguous bytes.
    if (a.rq_out != nullptr) {
        // Fused re-quantisation: 8-bit codes instead of fp32.  The workgroup's MT x NTP tile of codes goes through ONE byte
        // patch in LDS (behind the staging image and the channel sums: a.ptab_off), then every thread stores 16-byte
        // pieces of rows: NTP contiguous bytes per output channel instead of 4 x NTP.
        uint8_t *bp = smem + a.ptab_off;                                       // [MT][NTP]
        RqConst rqc = rq_setup(a);
        rqc.slow = __builtin_amdgcn_readfirstlane(rqc.slow);                   // the same in every lane: a branch, not a select
        rqc.chk = __builtin_amdgcn_readfirstlane(rqc.chk);
        bool bad = false;
        // wave-uniform: packed pairs (rq_fast2) when no check can fire, else element by element
        const bool fast = rq_fast_ok(rqc) && __builtin_amdgcn_ballot_w64(!rq_bounded(alpha, cst, bia, zwp)) == 0ull;
        auto body = [&](auto fast_tag, auto sx_tag) __attribute__((always_inline)) {
            constexpr bool FAST = decltype(fast_tag)::value, SXE = decltype(sx_tag)::value;
#pragma unroll
            for (int t = 0; t < NIW; ++t) {
                const int q0 = (wn + t * WN) * 32;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    uint32_t pk = 0;
                    if constexpr (FAST) {
#pragma unroll
                        for (int j = 0; j < 4; j += 2) {
                            v2f f = {(float)acc[t][4 * gq + j], (float)acc[t][4 * gq + j + 1]};
                            f = f + v2f{cst, cst};
                            if constexpr (SXE) {
                                const v2f sx2 = {(float)sxp[q0 + 8 * gq + 4 * h + j], (float)sxp[q0 + 8 * gq + 4 * h + j + 1]};
                                f = __builtin_elementwise_fma(v2f{-zwp, -zwp}, sx2, f);
                            }
                            const v2f r = rq_fast2(rqc, __builtin_elementwise_fma(v2f{alpha, alpha}, f, v2f{bia, bia}));
                            pk = __builtin_amdgcn_cvt_pk_u8_f32(r.x, j, pk);
                            pk = __builtin_amdgcn_cvt_pk_u8_f32(r.y, j + 1, pk);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float f = (float)acc[t][4 * gq + j] + cst;
                            if (need_sx) f = fmaf(-zwp, (float)sxp[q0 + 8 * gq + 4 * h + j], f);
                            pk = __builtin_amdgcn_cvt_pk_u8_f32(rq_value(rqc, fmaf(alpha, f, bia), bad), j, pk);
                        }
                    }
                    *reinterpret_cast<uint32_t *>(bp + (wm * 32 + col) * NTP + q0 + 8 * gq + 4 * h) = pk;
                }
            }
        };
        if (fast) { if (need_sx) body(std::true_type{}, std::true_type{}); else body(std::true_type{}, std::false_type{}); }
        else body(std::false_type{}, std::false_type{});
        // lanes of channel rows >= OC and pixels >= NT computed on padding: their codes are never stored, their range flags dropped
        if (oc >= a.OC) bad = false;
        __syncthreads();
        constexpr int PPR = NTP / 16;                                          // 16-byte pieces per row
        for (int e = tid; e < MT * PPR; e += THR) {
            const int row = e / PPR, px = 16 * (e - row * PPR);
            const int oc_r = ot * MT + row;
            if (oc_r < a.OC && px < NT) {
                const uint4 d4 = *reinterpret_cast<const uint4 *>(bp + row * NTP + px);
                uint8_t *dst = a.rq_out + ((int64_t)n * a.OC + oc_r) * P + p0 + px;
                if (px + 16 <= NT) {
                    __builtin_memcpy(dst, &d4, 16);                            // dword aligned (P % 4 == 0)
                } else {                                                       // NT % 4 == 0: whole dwords
                    const uint32_t dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (px + 4 * k < NT) __builtin_memcpy(dst + 4 * k, &dd[k], 4);
                }
            }
        }
        rq_report(a, bad);
        return;
    }
    float *out_w = a.out + ((int64_t)n * a.OC + ot * MT + wm * 32) * P + p0;   // wave-uniform
    float *patch = reinterpret_cast<float *>(smem) + wave * (32 * 36);         // staging LDS is free now
    const int rrow = lane >> 3, rq = lane & 7;                                 // read-back: row within 8, pixel quad
    const uint32_t voff = (uint32_t)rrow * (uint32_t)P + 4u * (uint32_t)rq;
#pragma unroll
    for (int t = 0; t < NIW; ++t) {
        const int q0 = (wn + t * WN) * 32;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float f = (float)acc[t][4 * gq + j] + cst;
                if (need_sx) f = fmaf(-zwp, (float)sxp[q0 + 8 * gq + 4 * h + j], f);
                v[j] = fmaf(alpha, f, bia);
            }
            *reinterpret_cast<float4 *>(patch + col * 36 + 8 * gq + 4 * h) = make_float4(v[0], v[1], v[2], v[3]);
        }
        // same wave wrote and reads: only the LDS counter has to drain
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
        const bool px_ok = q0 + 4 * rq < NT;
        if (q0 + 32 <= NT && ot * MT + wm * 32 + 32 <= a.OC) {   // wave-uniform: whole 32 x 32 tile inside, plain stores
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 o4 = *reinterpret_cast<const float4 *>(patch + (8 * i + rrow) * 36 + 4 * rq);
                *reinterpret_cast<float4 *>(out_w + (int64_t)(8 * i) * P + q0 + voff) = o4;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 8 * i + rrow;
                const float4 o4 = *reinterpret_cast<const float4 *>(patch + row * 36 + 4 * rq);
                if (px_ok && ot * MT + wm * 32 + row < a.OC)
                    *reinterpret_cast<float4 *>(out_w + (int64_t)(8 * i) * P + q0 + voff) = o4;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);   // reads done before the next tile overwrites the patch
    }
#ifdef QE_STAMP
    QE_ST(6);       // epilogue: conversions, patch round trips, stores issued
    __builtin_amdgcn_s_waitcnt(0x0070);   // vmcnt(0): stores acknowledged
    QE_ST(7);       // store drain
    if (a.dbg != nullptr && lane == 0) {
        unsigned long long *o = a.dbg + ((size_t)blockIdx.x * (WM * WN) + wave) * 10;
        for (int i = 0; i < 8; ++i) o[i] = st[i];
        o[8] = tprev - tstart;
        o[9] = tstart;
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// Flat 1x1 kernel for SMALL planes whose size is not a multiple of 4 (7x7 = 49 pixels: the last ResNet stage).
// Same GEMM, operand roles, weight path and store patch as conv_mfma_flat_kernel; what differs is the tile:
//   * a tile is GI whole images; image gi owns the pixel slots [gi * PS, gi * PS + P) of the LDS rows, PS = P rounded
//     up to 8 (49 -> 56; 4 images = 224 slots = 7 MFMA column tiles; slots P..PS-1 are never stored);
//   * a channel plane is P contiguous bytes at an arbitrary byte alignment (49 * k): it is fetched as 8-byte pieces
//     with byte-unaligned global_load_dwordx2 (probed: tools/probe_unaligned.hip), the last piece ENDING at the plane's
//     end and shifted down, so nothing is read past a plane; pieces land 8-byte aligned in LDS, which is all
//     ds_read_b64_tr_b8 needs.  14 piece loads per thread and 128-channel stage instead of the halo kernel's 32
//     unaligned dword loads + 8 byte transposes (its load issue alone was 38 % of a wave's life on 2048->512 @7x7);
//   * output rows are P floats at 4-byte alignment: global_store_dwordx4 at dword alignment (probed), scalar stores
//     for the last pixels of a plane.
// a.IWP carries PS, a.GI the images per tile.  Host guarantees CK * GI * ceil(P / 8) <= 16 * 256 pieces per stage.
// ---------------------------------------------------------------------------------------------
template <int NIW, int NS, bool WRAW, int NP8>
__global__ __launch_bounds__(MF_THREADS, 2) void conv_mfma_flatg_kernel(const MfmaArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int MT = 128;                     // 4 waves x 32 output channels
    constexpr int NTP = 32 * NIW;               // pixel slots per tile
    constexpr int RSTR = 32 * (NIW | 1);        // LDS row stride: odd multiple of 32 B
    constexpr int CK = 32 * NS;
    constexpr int GIM = 4;                      // image slots per tile the row map is built for
    constexpr int UPI = CK * NP8;               // 8-byte pieces of one image per stage
    constexpr int RPT = (UPI + MF_THREADS - 1) / MF_THREADS;   // piece rounds per thread (x GIM images each)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave;
    const int col = lane & 31, h = lane >> 5;
    const int P = a.H * a.W, PS = a.IWP, GI = a.GI;
    const int tail_r = P - 8 * (NP8 - 1);       // valid bytes of the last piece (1..8; host checks)

    int pt, ot;
    block_to_tile(a, pt, ot);
    if (pt >= a.n_pix_tiles) return;
    const int n0 = pt * GI;

    uint8_t *Xs = smem;
    constexpr int XS_BYTES = (CK * RSTR > 4 * 32 * 36 * 4) ? CK * RSTR : 4 * 32 * 36 * 4;
    int *sxp = reinterpret_cast<int *>(smem + XS_BYTES);

    const int oc = ot * MT + wm * 32 + col;
    const int occ = oc < a.OC ? oc : a.OC - 1;
    const float sw = a.w_per_tensor ? a.w_scale[0] : a.w_scale[occ];
    const float zwp = (a.w_per_tensor ? a.w_zero[0] : a.w_zero[occ]) - zero_shift(a.w_bits, a.w_sign);
    const float alpha = a.x_scale[0] * sw;
    const float bia = a.bias ? a.bias[occ] : 0.0f;
    const float zxp = a.x_zero[0] - zero_shift(a.x_bits, a.x_sign);

    // ---- staging (stage invariant): piece u = tid + 256 j <-> (channel c = u / NP8, piece k = u % NP8) of EVERY image
    // of the tile, so the lanes of a load instruction read consecutive 8-byte pieces of consecutive channel planes
    // (one contiguous span per image).  The last piece of a plane starts at P - 8 and is shifted down.
    int pcl[RPT], plds[RPT];
    uint32_t pgo[RPT];
    bool ptail[RPT];
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        const int u = tid + MF_THREADS * j;
        const int c = u / NP8, kk = u - c * NP8;
        const bool ok = u < UPI;
        pcl[j] = ok ? c : 0;
        ptail[j] = kk == NP8 - 1;
        pgo[j] = (uint32_t)(ptail[j] ? P - 8 : 8 * kk);
        plds[j] = ok ? c * RSTR + 8 * kk : -1;
    }
    const int tail_sh = 8 * (8 - tail_r);
    const uint8_t *xg = a.x + (int64_t)n0 * a.IC * P;
    const int64_t img_stride = (int64_t)a.IC * P;

    const int NGR = (a.IC + 15) >> 4;
    const uint8_t *w_lane = WRAW ? a.w_raw + (int64_t)occ * a.IC : nullptr;
    const int8_t *wt_base = a.wt + (int64_t)(ot * MT + wm * 32) * 16;
    const uint32_t wt_voff = (uint32_t)col * 16u;

    v16i acc[NIW];
#pragma unroll
    for (int t = 0; t < NIW; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0;
    }
    int swacc = 0;
    const int i16 = lane & 15;
    const int tr_base = (16 * h + (i16 >> 1)) * RSTR + 16 * ((lane >> 4) & 1) + 8 * (i16 & 1);

    uint2 d[RPT][GIM];
    auto issue_x = [&](int s) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const int cg = s * CK + pcl[j];
            const int cgc = cg < a.IC ? cg : a.IC - 1;
            const uint32_t off = (uint32_t)cgc * (uint32_t)P + pgo[j];
#pragma unroll
            for (int gi = 0; gi < GIM; ++gi) {
                const int gic = (gi < GI && n0 + gi < a.N) ? gi : 0;             // uniform
                __builtin_memcpy(&d[j][gi], xg + gic * img_stride + off, 8);     // byte-unaligned global_load_dwordx2
            }
        }
    };
    issue_x(0);

    const bool need_sx = __syncthreads_or((oc < a.OC && zwp != 0.0f) ? 1 : 0) != 0;
    if (need_sx) {
        for (int i = tid; i < NTP; i += MF_THREADS) sxp[i] = 0;
        __syncthreads();
    }

    auto stage_x = [&](int s) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
#pragma unroll
            for (int gi = 0; gi < GIM; ++gi) {
                unsigned long long v = ((unsigned long long)d[j][gi].y << 32) | d[j][gi].x;
                if (ptail[j]) v >>= tail_sh;                   // tail piece: its valid bytes come down to byte 0
                v ^= 0x8080808080808080ull;
                if (plds[j] >= 0 && gi < GI && n0 + gi < a.N) {
                    *reinterpret_cast<unsigned long long *>(Xs + plds[j] + gi * PS) = v;
                    if (need_sx && s * CK + pcl[j] < a.IC) {   // rare path (asymmetric weights)
                        const int pb = plds[j] - pcl[j] * RSTR + gi * PS;
                        for (int b = 0; b < 8; ++b) atomicAdd(&sxp[pb + b], (int)(int8_t)(v >> (8 * b)));
                    }
                }
            }
        }
    };
    auto stage = [&](int s, auto prefetch) __attribute__((always_inline)) {
        v4i wf[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int icg = 2 * (s * NS + k) + h;
            const int icgc = icg < NGR ? icg : NGR - 1;
            v4i f;
            if constexpr (WRAW) {
                __builtin_memcpy(&f, w_lane + icgc * 16, 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) f[j] ^= (int)0x80808080;
            } else {
                f = *reinterpret_cast<const v4i *>(wt_base + (int64_t)icgc * a.OCP * 16 + wt_voff);
            }
            if (icg >= NGR) f = v4i{0, 0, 0, 0};
            wf[k] = f;
        }
        stage_x(s);
        __syncthreads();
        if constexpr (decltype(prefetch)::value) issue_x(s + 1);
#pragma unroll
        for (int k = 0; k < NS; ++k) {
#pragma unroll
            for (int j = 0; j < 4; ++j) swacc = __builtin_amdgcn_sdot4(wf[k][j], 0x01010101, swacc, false);
#pragma unroll
            for (int t = 0; t < NIW; ++t) {
                const uint8_t *src = Xs + tr_base + (k * 32) * RSTR + t * 32;
                const v2i lo = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3))) *)(src));
                const v2i hi = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3))) *)(src + 8 * RSTR));
                const v4i xf = {lo[0], lo[1], hi[0], hi[1]};
                acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xf, wf[k], acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
    };

    const int n_stages = (a.IC + CK - 1) / CK;
    for (int s = 0; s < n_stages - 1; ++s) stage(s, std::true_type{});
    stage(n_stages - 1, std::false_type{});

    // ---- epilogue: lane = output channel, 4 consecutive registers = 4 consecutive slots; per-wave LDS patch ----
    const int sw_sum = swacc + __shfl_xor(swacc, 32);
    const float cst = fmaf((float)a.IC * zxp, zwp, -zxp * (float)sw_sum);
    if (a.rq_out != nullptr) {
        // Fused re-quantisation (see conv_mfma_flat_kernel).  Byte patch in OUTPUT order [image gi][oc row][P]: the MT planes
        // of one image are one contiguous run of MT * P bytes of the output tensor, stored as 16-byte pieces at byte alignment
        // (49-byte planes; unaligned global_store_dwordx4: tools/probe_unaligned.hip).
        uint8_t *bp = smem + a.ptab_off;
        const RqConst rqc = rq_setup(a);
        bool bad = false;
#pragma unroll
        for (int t = 0; t < NIW; ++t) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int m = t * 32 + 8 * gq + 4 * h;          // 4 consecutive slots: image gi, pixels p .. p + 3
                const int gi = m / PS, p = m - gi * PS;
                uint8_t *dst = bp + ((gi < GI ? gi : 0) * MT + wm * 32 + col) * P + p;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float f = (float)acc[t][4 * gq + j] + cst;
                    if (need_sx) f = fmaf(-zwp, (float)sxp[m + j], f);
                    bool b1 = false;
                    const unsigned c = (unsigned)rq_value(rqc, fmaf(alpha, f, bia), b1);
                    if (gi < GI && p + j < P) { dst[j] = (uint8_t)c; bad |= b1 && oc < a.OC && n0 + gi < a.N; }
                }
            }
        }
        __syncthreads();
        const int rows = min(MT, a.OC - ot * MT);           // channel rows of this tile that exist
        const int run = rows * P;                           // bytes per image
        const int ppi = (run + 15) >> 4;                    // 16-byte pieces per image
        for (int e = tid; e < GI * ppi; e += MF_THREADS) {
            const int gi = e / ppi, b = 16 * (e - gi * ppi);
            if (n0 + gi >= a.N) continue;
            const uint8_t *src = bp + gi * MT * P + b;
            uint8_t *dst = a.rq_out + ((int64_t)(n0 + gi) * a.OC + ot * MT) * P + b;
            if (b + 16 <= run) {
                const uint4 d4 = *reinterpret_cast<const uint4 *>(src);   // LDS side is 16-byte aligned (MT * P % 16 == 0)
                __builtin_memcpy(dst, &d4, 16);
            } else {
                for (int k = 0; b + k < run; ++k) dst[k] = src[k];
            }
        }
        rq_report(a, bad);
        return;
    }
    float *patch = reinterpret_cast<float *>(smem) + wave * (32 * 36);
    const int rrow = lane >> 3, rq = lane & 7;
    const int row_oc = ot * MT + wm * 32;               // first output channel of this wave's strip
#pragma unroll
    for (int t = 0; t < NIW; ++t) {
        const int q0 = t * 32;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float f = (float)acc[t][4 * gq + j] + cst;
                if (need_sx) f = fmaf(-zwp, (float)sxp[q0 + 8 * gq + 4 * h + j], f);
                v[j] = fmaf(alpha, f, bia);
            }
            *reinterpret_cast<float4 *>(patch + col * 36 + 8 * gq + 4 * h) = make_float4(v[0], v[1], v[2], v[3]);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): same wave wrote and reads
        // this lane's 4 slots of the tile: image gi, pixels p .. p+3 (PS % 4 == 0: never across images)
        const int m = q0 + 4 * rq;
        const int gi = m / PS, p = m - gi * PS;
        const bool img_ok = gi < GI && n0 + gi < a.N;
        float *orow = a.out + ((int64_t)(n0 + (img_ok ? gi : 0)) * a.OC + row_oc) * P + p;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 8 * i + rrow;
            const float4 o4 = *reinterpret_cast<const float4 *>(patch + row * 36 + 4 * rq);
            if (img_ok && row_oc + row < a.OC) {
                float *dst = orow + (int64_t)row * P;
                if (p + 4 <= P) {
                    __builtin_memcpy(dst, &o4, 16);       // dword-aligned global_store_dwordx4
                } else {
                    if (p < P) dst[0] = o4.x;
                    if (p + 1 < P) dst[1] = o4.y;
                    if (p + 2 < P) dst[2] = o4.z;
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
}

// launchers, one translation unit per wave layout (qe_conv_mfma_i*.hip)
void launch_mfma_cfg0(const MfmaArgs &a, int niw, int ns, int KK, bool x8, unsigned blocks, size_t lds, hipStream_t s);
void launch_mfma_cfg1(const MfmaArgs &a, int niw, int ns, int KK, bool x8, unsigned blocks, size_t lds, hipStream_t s);
void launch_mfma_cfg2(const MfmaArgs &a, int niw, int ns, int KK, bool x8, unsigned blocks, size_t lds, hipStream_t s);
void launch_mfma_smallic(const MfmaArgs &a, int cfg, int niw, unsigned blocks, size_t lds, hipStream_t s);
void launch_mfma_ws(const MfmaArgs &a, int niw, int split, unsigned blocks, size_t lds, hipStream_t s);
void launch_mfma_sm2(const MfmaArgs &a, int wms, int split, unsigned blocks, size_t lds, hipStream_t s);
void launch_mfma_flatg(const MfmaArgs &a, int ns, bool wraw, unsigned blocks, size_t lds, hipStream_t s);
void launch_mfma_flat(const MfmaArgs &a, int cfg, int niw, int ns, bool wraw, bool s2, unsigned blocks, size_t lds, hipStream_t s);
void launch_mfma_flat_x4(const MfmaArgs &a, int niw, int ns, unsigned blocks, size_t lds, hipStream_t s);

#define QE_MFMA_K(WM, WN, NIW, KKT, X8, NS)                                                                                     \
    do {                                                                                                                        \
        if (a.rq_out != nullptr && a.rq_patch && (WM) == 4 && (NIW) == 7 && (KKT) == 9 && (X8) && (NS) == 1)                    \
            hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, NIW, KKT, X8, NS, true, (WM) == 4 && (NIW) == 7 && (KKT) == 9 && (X8) && (NS) == 1>), dim3(blocks), dim3(MF_THREADS), lds, s, a); \
        else if (a.rq_out != nullptr)                                                                                           \
            hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, NIW, KKT, X8, NS, true>), dim3(blocks), dim3(MF_THREADS), lds, s, a);  \
        else                                                                                                                    \
            hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, NIW, KKT, X8, NS, false>), dim3(blocks), dim3(MF_THREADS), lds, s, a); \
    } while (0)

// 1x1 convolutions get the multi-chunk stages (ns = 1, 2, 4); sub-8-bit activations only ns = 1
#define QE_MFMA_LAUNCH(WM, WN, NIW)                                                   \
    do {                                                                              \
        if (KK == 1) {                                                                \
            if (!x8)           QE_MFMA_K(WM, WN, NIW, 1, false, 1);                   \
            else if (ns == 4)  QE_MFMA_K(WM, WN, NIW, 1, true, 4);                    \
            else if (ns == 2)  QE_MFMA_K(WM, WN, NIW, 1, true, 2);                    \
            else               QE_MFMA_K(WM, WN, NIW, 1, true, 1);                    \
        } else if (KK == 9 && a.KW == 3) {                                            \
            if (x8) QE_MFMA_K(WM, WN, NIW, 9, true, 1);                               \
            else    QE_MFMA_K(WM, WN, NIW, 9, false, 1);                              \
        } else {                                                                      \
            if (x8) QE_MFMA_K(WM, WN, NIW, 0, true, 1);                               \
            else    QE_MFMA_K(WM, WN, NIW, 0, false, 1);                              \
        }                                                                             \
    } while (0)

}  // namespace qe
