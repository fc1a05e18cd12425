// qe_conv_pwr.hip -- 1x1 / stride 1 / pad 0 quantconv2d with the activation tile RESIDENT in LDS ("pwr": pointwise, resident).
//
// Replaces the per-element loop of quantconv2d_cuda_kernel (engine/kernels/functions/quantconv2d.cu:78-141) for 8-bit x 8-bit
// 1x1 layers whose whole channel depth of one pixel tile fits the LDS (IC = 64 | 128 | 256): the expansion layers of a
// bottleneck network (64->256 @56x56, 128->512 @28x28, 256->1024 @14x14: 14 of the 53 ResNet-50 convolutions, 1.5 of the
// 4.2 ms of the batch-256 step).  Same arithmetic as the flat kernels (qe_conv_mfma_kernel.hpp, qe_conv_flatd.hip):
//   out[n,oc,p] = bias[oc] + sx sw[oc] ( S_aw - zw' S_x - zx' S_w + IC zx' zw' ),  a = u ^ 0x80, S_aw exact in int32 on
//   v_mfma_i32_32x32x32_i8.
//
// Why another kernel.  These layers write 4 x OC bytes per pixel and read IC: they are bound by the store stream.  The flat
// kernels give every (pixel tile, 128 output channels) pair its own workgroup: each of the OC/128 workgroups of a tile
// fetches the tile again, waits one memory round trip per 64-channel stage with nothing else in flight, and only then
// stores -- and a CU's loads queue behind the stores of its other resident workgroup, so the load phases and the store
// phases of a CU add up (F = A + B on every layer).  Here
//   * a workgroup owns a pixel tile for ALL (or 1/OCSPLIT of) the output channels: the tile's IC x T bytes arrive ONCE, by
//     LDS-DMA (global_load_lds_dwordx4, every piece of the tile requested before the first is waited for), are recoded
//     (u ^ 0x80) once in place, and stay;
//   * after ONE barrier the waves never synchronise again: wave w walks the 32-channel strips w, w + WAVES, ...; per strip
//     it multiplies (weights straight from the packed OIHW rows, L2 -> VGPR, requested one strip ahead and BEFORE the
//     previous strip's stores, so the in-order vmcnt never parks them behind a store acknowledgement), converts and
//     stores.  The waves of a CU drift apart, so its store stream never pauses for a load phase;
//   * operand roles: A = weights (rows = output channel), B = activations (columns = pixel, ds_read_b64_tr_b8 from the
//     native [channel][pixel] image).  A register quad of an accumulator tile is then 8 consecutive output channels x 32
//     pixels: the wave turns 8 channels x the WHOLE tile width through a private LDS patch and stores it as 16-byte pieces
//     of 896-byte row runs -- or, when the tile is a whole plane (14x14), as ONE contiguous, line-aligned 6272-byte run
//     (8 rows x 784 B = 49 lines), which the 8-rows-x-128-B pieces of the flat kernels cannot give on 784-byte rows
//     (store-only probes: 3.6 TB/s against 5.1-5.6, profiles/r02c_probe_store_occ.txt).
#include "qe_conv_mfma_kernel.hpp"

#include <cstdlib>
#include <utility>

namespace qe {

struct PwrArgs {
    const uint8_t *x;          // [N][IC][P] stored codes, 8-bit
    const uint8_t *w;          // [OC][IC] stored codes, 8-bit
    const float *x_scale, *x_zero, *w_scale, *w_zero, *bias;
    int x_sign, w_sign, w_per_tensor;
    float *out;                // [N][OC][P] fp32
    int N, IC, OC, P;
    int tiles_per_image, n_pix_tiles, chunk;
    int n_groups;              // output-channel groups a pixel tile is split over (workgroups per tile)
    int strips_per_group;      // 32-channel strips of one group
    unsigned long long *dbg;   // diagnostic builds (-DQE_STAMP) only: per-wave phase cycle sums
    int W_in, PIN, OW;         // stride-2 form: input row length, input plane size, output row length (P = output plane)
    // fused re-quantisation (RQ instances): the 8-bit code of the consumer's quantiser instead of fp32, fields as in MfmaArgs
    uint8_t *rq_out;
    const float *rq_scale, *rq_zero;
    float rq_qmin, rq_qmax, rq_lo, rq_hi;
    unsigned rq_offset;
    int32_t *rq_status;
};

// NT column tiles of 32 pixel slots (odd), of which the first TW pixels are real: every tile of a launch has the same
// width (the host picks TW | P), so the number of stores per strip is a compile-time constant (see wait_w below).
template <int NT, int WAVES, int KS, int TW> struct PwrGeom {
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int IC = 32 * KS;
    static constexpr int RS = 32 * NT;                       // LDS bytes per channel row (NT odd: conflict-free transposed reads)
    static constexpr int NTP = 32 * NT;                      // pixel slots per tile
    static constexpr int XBYTES = IC * RS;
    static constexpr int XINSTR = XBYTES / 1024;             // wave-level DMA instructions of the tile
    static constexpr int PXW = (XINSTR + WAVES - 1) / WAVES; // ... per wave
    static constexpr int TAB = WAVES * 4 * 32 * 4;           // per wave: alpha, cst, bias, zw' of its strip's 32 channels
    static constexpr int PATCH = 8 * TW * 4;                 // per wave: 8 channel rows x the tile width, fp32
    static constexpr int PPR = TW / 4;                       // 16-byte pieces per patch row
    static constexpr int NRB = (8 * PPR + 63) / 64;          // read-back / store rounds per register quad
    static constexpr int PPRB = (TW + 15) / 16;              // RQ: 16-byte pieces per row of codes (the last one shifted back to end at the row's end)
    static constexpr int NRQ = (32 * PPRB + 63) / 64;        // RQ: store rounds per STRIP (32 rows of TW bytes through the same patch space)
    static constexpr int LDS = XBYTES + TAB + WAVES * PATCH;
};

// (Tried and removed: every accumulator register straight out as one buffer_store_dword -- two 128-byte row segments per
// instruction, no LDS round trip, 3 vector instructions per element.  In-kernel stamps: under load the epilogues' issue
// time went UP, 50-57 k -> 78 k cycles per wave on 256 -> 1024 @14x14: a back-pressured store costs its 500+ cycles per
// INSTRUCTION whatever its width, so the 1 KiB stores of the patch form are worth their LDS round trip.)
// S2: 1x1 / stride 2 / pad 0 (the downsample branch of a stage's first block): the tile is TW / OW output rows; its
// pixels are the even columns of the even input rows, fetched ONCE per tile as 8-byte pieces of those rows (registers,
// v_perm keeps the even bytes, ds_write_b32) instead of once per 128 output channels by the flat kernel's strided staging.
// RQ: fused re-quantisation (qe_quantconv2d_requant_prepared).  The operand roles are swapped (A = activations, B = weights:
// the transposed accumulator), so a lane owns ONE output channel and 4 consecutive pixels per register quad: its channel
// constants are its own registers (no table), four codes pack into a dword (v_cvt_pk_u8_f32) and the strip's 32 rows x TW
// bytes go through the wave's patch once: NRQ = 7 store instructions per strip instead of 28.
template <int NT, int WAVES, int KS, int TW, bool S2 = false, bool RQ = false>
__global__ __launch_bounds__(64 * WAVES, 2) void conv_pwr_kernel(const PwrArgs a)
{
    using G = PwrGeom<NT, WAVES, KS, TW>;
    static_assert(TW % 4 == 0 && TW <= 32 * NT && TW > 32 * (NT - 1), "tile width");
    constexpr int RS = G::RS, PXW = G::PXW;
    static_assert((NT & 1) == 1, "row stride must be an odd multiple of 32 B");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, h = lane >> 5;
    const int P = a.P;
#ifdef QE_STAMP
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = qe_stamp();
    const unsigned long long tstart = tprev;
#endif

    // ---- tile decode: XCD-aware block map (the scheme of block_to_tile; "oc tile" = channel group here) ----------
    int pt, og;
    {
        const int bid = blockIdx.x;
        const int idx = bid >> 3;
        const int j = idx / a.n_groups;
        og = idx - j * a.n_groups;
        const int c = j / a.chunk;
        pt = (c * 8 + (bid & 7)) * a.chunk + (j - c * a.chunk);
    }
    if (pt >= a.n_pix_tiles) return;
    const int n0 = pt / a.tiles_per_image;
    const int p0 = (pt - n0 * a.tiles_per_image) * TW;        // P % TW == 0: every tile is TW pixels wide

    uint8_t *Xs = smem;
    float *tab = reinterpret_cast<float *>(smem + G::XBYTES) + wave * 128;          // [4][32]
    float *patch = reinterpret_cast<float *>(smem + G::XBYTES + G::TAB) + wave * (8 * TW);

    // ---- weights + per-channel constants of a strip: lane (row oc0 + col, k half h) reads 16 contiguous bytes of its
    // packed OIHW row per 32-channel step.  Requested one strip ahead and BEFORE the previous strip's stores; the
    // constants go first so that the (in-order) wait for the weights covers them too.
    const int strip0 = og * a.strips_per_group + wave;        // this wave's strips: strip0, strip0 + WAVES, ...
    const int n_my = (a.strips_per_group - wave + WAVES - 1) / WAVES;   // <= 0: nothing to compute (still stages X)
    // The loads are inline asm: hipcc cannot count the stores between them and their use (they sit behind wave-uniform
    // branches), so its own wait would be vmcnt(7..0) -- i.e. for the acknowledgement of every store of the previous strip.
    // Form (ii) of the guide's section 5.7: "=v" loads, then ONE wait statement naming every destination "+v".
    v4i wf[KS];
    float c_sw, c_zw, c_bi;
    auto load_w = [&](int strip) __attribute__((always_inline)) {
        const int oc = strip * 32 + col;
        const float *psw = a.w_scale + (a.w_per_tensor ? 0 : oc);
        const float *pzw = a.w_zero + (a.w_per_tensor ? 0 : oc);
        const float *pbi = a.bias ? a.bias + oc : psw;          // no bias: any valid address, the value is dropped
        const uint8_t *wl = a.w + (int64_t)oc * G::IC + 16 * h;
        asm volatile("global_load_dword %0, %3, off\n\tglobal_load_dword %1, %4, off\n\tglobal_load_dword %2, %5, off"
                     : "=&v"(c_sw), "=&v"(c_zw), "=&v"(c_bi) : "v"(psw), "v"(pzw), "v"(pbi) : "memory");
#define QE_PWR_LW(K, OFF) if constexpr (KS > K) asm volatile("global_load_dwordx4 %0, %1, off offset:" #OFF : "=v"(wf[K]) : "v"(wl) : "memory")
        QE_PWR_LW(0, 0); QE_PWR_LW(1, 32); QE_PWR_LW(2, 64); QE_PWR_LW(3, 96);
        QE_PWR_LW(4, 128); QE_PWR_LW(5, 160); QE_PWR_LW(6, 192); QE_PWR_LW(7, 224);
#undef QE_PWR_LW
    };
    // wait until at most N vector-memory operations issued AFTER the weight requests are outstanding.  N is a compile-time
    // constant and there is ONE statement per call site: a run-time switch over several such statements made hipcc copy the
    // destination registers in front of the wait (i.e. before the data had to be there).
    auto wait_w = [&](auto n_tag) __attribute__((always_inline)) {
        constexpr int N = decltype(n_tag)::value;
        if constexpr (KS == 2) asm volatile("s_waitcnt vmcnt(%5)" : "+v"(c_sw), "+v"(c_zw), "+v"(c_bi), "+v"(wf[0]), "+v"(wf[1]) : "i"(N) : "memory");
        else if constexpr (KS == 4) asm volatile("s_waitcnt vmcnt(%7)" : "+v"(c_sw), "+v"(c_zw), "+v"(c_bi), "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]) : "i"(N) : "memory");
        else asm volatile("s_waitcnt vmcnt(%11)" : "+v"(c_sw), "+v"(c_zw), "+v"(c_bi), "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]), "+v"(wf[4]), "+v"(wf[5]), "+v"(wf[6]), "+v"(wf[7]) : "i"(N) : "memory");
        if (!a.bias) c_bi = 0.0f;
    };
    if (n_my > 0) load_w(strip0);

    const int n_xi = (G::XINSTR % WAVES == 0 || wave < G::XINSTR % WAVES) ? PXW : PXW - 1;
    int fix_i = -1;
    uint32_t tail_word = 0;
    if constexpr (S2) {
        // unit u = tid + THREADS * i <-> (channel c = u / RT, output row r = u % RT of the tile); its input row is W_in bytes =
        // W_in / 8 pieces, each giving one dword of 4 output pixels.  Every load stays inside its row.
        constexpr int THREADS = 64 * WAVES;
        constexpr int MAXP = 8;                               // pieces per row (W_in <= 64)
        const int RT = TW / a.OW;
        const int NU = G::IC * RT;
        const int np = a.W_in >> 3;
        const int r0 = p0 / a.OW;
        const uint8_t *xin = a.x + (int64_t)n0 * G::IC * a.PIN;
        constexpr int UPT = (G::IC * 8 + THREADS - 1) / THREADS;   // RT <= 8
#pragma unroll
        for (int i0 = 0; i0 < UPT; i0 += 2) {                 // two units' loads in flight per thread (2 x 16 VGPRs)
            uint2 d[2][MAXP];
            int cc[2], rr[2];
            bool live[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int u = tid + THREADS * (i0 + q);
                live[q] = (i0 + q) < UPT && u < NU;
                const int uc = live[q] ? u : 0;
                cc[q] = uc / RT; rr[q] = uc - cc[q] * RT;
                const uint8_t *row = xin + (int64_t)cc[q] * a.PIN + (int64_t)(2 * (r0 + rr[q])) * a.W_in;
#pragma unroll
                for (int j = 0; j < MAXP; ++j) {
                    const int jc = j < np ? j : np - 1;       // unconditional loads (clamped), unused pieces dropped below
                    __builtin_memcpy(&d[q][j], row + 8 * jc, 8);
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                uint32_t *dst = reinterpret_cast<uint32_t *>(Xs + cc[q] * RS + rr[q] * a.OW);
#pragma unroll
                for (int j = 0; j < MAXP; ++j)
                    if (live[q] && j < np) dst[j] = __builtin_amdgcn_perm(d[q][j].y, d[q][j].x, 0x06040200u) ^ 0x80808080u;
            }
        }
    } else {
    // ---- the tile: IC x RS bytes by LDS-DMA, slot e = 64 * (wave-instruction q) + lane, q = wave, wave + WAVES, ... ----
    // slot (channel c, j) holds plane bytes [p0 + 16 j, + 16) of channel c.  Only slots of the tensor's LAST plane can
    // reach past its end: pure-garbage slots fetch the tensor's last 16 bytes instead, the one partly valid slot
    // (P % 16 == 4: its 4 valid bytes are the tensor's last dword) is left out of the DMA and written by its lane.
    const int64_t x_total = (int64_t)a.N * G::IC * P;
    const int64_t x_last16 = x_total - 16;
    tail_word = *reinterpret_cast<const uint32_t *>(a.x + x_total - 4);   // x 4-byte aligned, P % 4 == 0
#pragma unroll
    for (int i = 0; i < PXW; ++i) {
        const int e = 64 * (wave + WAVES * i) + lane;
        const int c = e / (RS / 16);
        const int j = e - c * (RS / 16);
        int64_t src = ((int64_t)n0 * G::IC + c) * P + p0 + 16 * j;
        bool skip = false;
        if (src > x_last16) {
            if (src < x_total && i < n_xi) { skip = true; fix_i = i; }   // starts inside the tensor, ends past it
            else src = x_last16;
        }
        if (i < n_xi && !skip)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.x + src),
                                             (__attribute__((address_space(3))) void *)(Xs + 1024 * (wave + WAVES * i)), 16, 0, 0);
    }

    }

    // ---- per-lane constants of the read-back: piece f = 64 k + lane of the 8 x TW patch -> (row, 16-byte piece) ----
    constexpr int NRB = G::NRB;
    uint32_t rb_off[NRB];                                     // element offset inside the 8-row output block
#pragma unroll
    for (int k = 0; k < NRB; ++k) {
        const int f = 64 * k + lane;
        const int row = f / G::PPR, pc = f - row * G::PPR;
        rb_off[k] = f < 8 * G::PPR ? (uint32_t)row * (uint32_t)P + 4u * (uint32_t)pc : 0u;
    }

    // RQ: piece f = 64 k + lane of the strip's 32 x TW bytes of codes.  Tile = whole plane: the 32 rows are ONE contiguous
    // run of the output, copied flat as aligned pieces; else row f / PPRB, piece f % PPRB, the last piece of a row shifted
    // back to end at the row's end (dword-aligned 16-byte stores; the bytes two pieces share hold the same codes).
    uint32_t rq_lds[RQ ? G::NRQ : 1], rq_glb[RQ ? G::NRQ : 1];
    bool rq_live[RQ ? G::NRQ : 1];
    if constexpr (RQ) {
#pragma unroll
        for (int k = 0; k < G::NRQ; ++k) {
            const int f = 64 * k + lane;
            if (P == TW) {
                rq_live[k] = 16 * f < 32 * TW;
                rq_lds[k] = rq_glb[k] = rq_live[k] ? 16u * (uint32_t)f : 0u;
            } else {
                const int row = f / G::PPRB, pc = f - row * G::PPRB;
                const int boff = 16 * pc < TW - 16 ? 16 * pc : TW - 16;
                rq_live[k] = f < 32 * G::PPRB;
                rq_lds[k] = rq_live[k] ? (uint32_t)(row * TW + boff) : 0u;
                rq_glb[k] = rq_live[k] ? (uint32_t)row * (uint32_t)P + (uint32_t)boff : 0u;
            }
        }
    }
    RqConst rqc;
    bool bad = false, rq_fast_u = false;
    if constexpr (RQ) {
        rqc = rq_setup(a);
        rqc.slow = __builtin_amdgcn_readfirstlane(rqc.slow);      // the same in every lane: say so (or every use becomes a select)
        rqc.chk = __builtin_amdgcn_readfirstlane(rqc.chk);
        rq_fast_u = rq_fast_ok(rqc);
    }
    const float zxp = a.x_zero[0] - (a.x_sign ? 0.0f : 128.0f);
    const float sx = a.x_scale[0];
    const float zw_shift = a.w_sign ? 0.0f : 128.0f;
    QE_ST(0);   // prologue: requests issued
    // ---- recode the pieces this lane fetched (u ^ 0x80: signed q, or unsigned q - 128), once for all strips --------
    if (n_my > 0) wait_w(std::integral_constant<int, 0>{}); else __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): tile pieces and first weights landed
#pragma unroll
    for (int i = 0; i < PXW; ++i) {
        if (!S2 && i < n_xi) {
            uint4 *slot = reinterpret_cast<uint4 *>(Xs + 1024 * (wave + WAVES * i) + 16 * lane);
            uint4 v = *slot;
            if (i == fix_i) v.x = tail_word;
            v.x ^= 0x80808080u; v.y ^= 0x80808080u; v.z ^= 0x80808080u; v.w ^= 0x80808080u;
            *slot = v;
        }
    }
    QE_ST(1);   // tile landed + recode pass
    __syncthreads();                                          // the only barrier: the tile is complete
    QE_ST(2);   // barrier
    if (n_my <= 0) return;

    // transposed-read base of this lane: row (16 h + i16 / 2) of a 32-channel step, 16-pixel group (lane >> 4) & 1
    const int i16 = lane & 15;
    const uint8_t *tr_base = Xs + (16 * h + (i16 >> 1)) * RS + 16 * ((lane >> 4) & 1) + 8 * (i16 & 1);

    float sxv[NT];                                            // S_x of this lane's pixel in column tile t (when needed)
#pragma unroll
    for (int t = 0; t < NT; ++t) sxv[t] = 0.0f;

    v16i acc[NT];
    int swacc;
    auto mma_strip = [&](auto sx_tag) __attribute__((always_inline)) {
        constexpr bool SX = decltype(sx_tag)::value;
        int sxacc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            sxacc[t] = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0;
        }
        swacc = 0;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            v4i wk = wf[k];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                wk[j] ^= (int)0x80808080;
                swacc = __builtin_amdgcn_sdot4(wk[j], 0x01010101, swacc, false);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const uint8_t *src = tr_base + (k * 32) * RS + t * 32;
                const v2i lo = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3))) *)(src));
                const v2i hi = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3))) *)(src + 8 * RS));
                const v4i xf = {lo[0], lo[1], hi[0], hi[1]};
                if constexpr (SX) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) sxacc[t] = __builtin_amdgcn_sdot4(xf[j], 0x01010101, sxacc[t], false);
                }
                if constexpr (RQ) acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xf, wk, acc[t], 0, 0, 0);
                else acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wk, xf, acc[t], 0, 0, 0);
            }
        }
        if constexpr (SX) {
#pragma unroll
            for (int t = 0; t < NT; ++t) sxv[t] = (float)(sxacc[t] + __shfl_xor(sxacc[t], 32));
        }
    };

    // epilogue of the strip whose sums sit in acc / swacc; (e_sw, e_zw, e_bi) = its lane's channel constants.
    // next >= 0: the next strip's weights are requested in front of this strip's N_YOUNGER stores (vmcnt is a 6-bit in-order
    // counter: the wait for the weights names the stores issued behind them).
    constexpr int N_YOUNGER = RQ ? G::NRQ : 4 * NRB;
    static_assert(N_YOUNGER <= 63, "vmcnt is a 6-bit counter");
    auto epilogue = [&](int strip, float e_sw, float e_zw, float e_bi, bool need_sx) __attribute__((always_inline)) {
        const int oc0 = strip * 32;
        if constexpr (RQ) {
            // lane col owns channel oc0 + col (its own constants); register 4 gq + j of column tile t = pixel 32 t + 8 gq + 4 h + j.
            // The fp32 value is computed exactly as in the fp32 form below, then quantised as quantize_pack would.
            const float zwp = e_zw - zw_shift;
            const int sw_sum = swacc + __shfl_xor(swacc, 32);
            const float cst = fmaf((float)G::IC * zxp, zwp, -zxp * (float)sw_sum);
            const float alpha = sx * e_sw;
            uint8_t *bp = reinterpret_cast<uint8_t *>(patch);                  // [32][TW] codes
            // wave-uniform choice (a real branch: left to itself hipcc evaluates both forms for every element and selects):
            // packed pairs when nothing can overflow or trip the range flag, else element by element with every check
            const bool fast = rq_fast_u && __builtin_amdgcn_ballot_w64(!rq_bounded(alpha, cst, e_bi, zwp)) == 0ull;
            auto body = [&](auto fast_tag, auto sx_tag) __attribute__((always_inline)) {
                constexpr bool FAST = decltype(fast_tag)::value, SXE = decltype(sx_tag)::value;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int px = 32 * t + 8 * gq + 4 * h;                // first of the 4 pixels; TW % 4 == 0: all four are real or none
                        const bool real = 32 * t + 32 <= TW || px < TW;        // first term compile-time
                        uint32_t pk = 0;
                        if constexpr (FAST) {
#pragma unroll
                            for (int j = 0; j < 4; j += 2) {
                                v2f f = {(float)acc[t][4 * gq + j], (float)acc[t][4 * gq + j + 1]};
                                f = f + v2f{cst, cst};
                                if constexpr (SXE) {
                                    const v2f sxp = {__shfl(sxv[t], 8 * gq + 4 * h + j), __shfl(sxv[t], 8 * gq + 4 * h + j + 1)};
                                    f = __builtin_elementwise_fma(v2f{-zwp, -zwp}, sxp, f);
                                }
                                const v2f r = rq_fast2(rqc, __builtin_elementwise_fma(v2f{alpha, alpha}, f, v2f{e_bi, e_bi}));
                                pk = __builtin_amdgcn_cvt_pk_u8_f32(r.x, j, pk);
                                pk = __builtin_amdgcn_cvt_pk_u8_f32(r.y, j + 1, pk);
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                float f = (float)acc[t][4 * gq + j] + cst;
                                if (need_sx) f = fmaf(-zwp, __shfl(sxv[t], 8 * gq + 4 * h + j), f);
                                bool b = false;
                                const float r = rq_value(rqc, fmaf(alpha, f, e_bi), b);
                                bad |= b && real;                              // padding slots hold whatever followed the tile
                                pk = __builtin_amdgcn_cvt_pk_u8_f32(r, j, pk);
                            }
                        }
                        if (real) *reinterpret_cast<uint32_t *>(bp + col * TW + px) = pk;
                    }
                }
            };
            if (fast) { if (need_sx) body(std::true_type{}, std::true_type{}); else body(std::true_type{}, std::false_type{}); }
            else body(std::false_type{}, std::false_type{});
            uint8_t *out_q = a.rq_out + ((int64_t)n0 * a.OC + oc0) * P + p0;   // wave-uniform
#pragma unroll
            for (int k = 0; k < G::NRQ; ++k) {
                uint4 d4;
                const uint32_t *src = reinterpret_cast<const uint32_t *>(bp + rq_lds[k]);   // dword aligned
                d4.x = src[0]; d4.y = src[1]; d4.z = src[2]; d4.w = src[3];
                if (rq_live[k]) __builtin_memcpy(out_q + rq_glb[k], &d4, 16);
            }
            return;
        }
        {
            // lane col owns channel oc0 + col; the accumulator rows read the constants back from LDS
            const float zwp = e_zw - zw_shift;
            const int sw_sum = swacc + __shfl_xor(swacc, 32);
            const float cst = fmaf((float)G::IC * zxp, zwp, -zxp * (float)sw_sum);
            if (h == 0) {
                tab[col] = sx * e_sw;
                tab[32 + col] = cst;
                tab[64 + col] = e_bi;
                tab[96 + col] = zwp;
            }
        }
        float *out_s = a.out + ((int64_t)n0 * a.OC + oc0) * P + p0;      // wave-uniform
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            // rows 8 gq + 4 h + j, j = 0..3 of the strip <-> registers 4 gq + j
            const float4 al = *reinterpret_cast<const float4 *>(tab + 8 * gq + 4 * h);
            const float4 cs = *reinterpret_cast<const float4 *>(tab + 32 + 8 * gq + 4 * h);
            const float4 bi = *reinterpret_cast<const float4 *>(tab + 64 + 8 * gq + 4 * h);
            const float alv[4] = {al.x, al.y, al.z, al.w}, csv[4] = {cs.x, cs.y, cs.z, cs.w};
            const float biv[4] = {bi.x, bi.y, bi.z, bi.w};
            float zwv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (need_sx) {
                const float4 zw = *reinterpret_cast<const float4 *>(tab + 96 + 8 * gq + 4 * h);
                zwv[0] = zw.x; zwv[1] = zw.y; zwv[2] = zw.z; zwv[3] = zw.w;
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int px = 32 * t + col;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // the flat kernels' operation order exactly (their fused re-quantising epilogue must give the codes
                    // of quantize_pack on THIS kernel's fp32 output bit for bit: tests/test_requant_gpu.py)
                    float f = (float)acc[t][4 * gq + j] + csv[j];
                    if (need_sx) f = fmaf(-zwv[j], sxv[t], f);
                    v[j] = fmaf(alv[j], f, biv[j]);
                }
                if (32 * t + 32 <= TW || px < TW) {           // first term compile-time: only the last column tile is masked
#pragma unroll
                    for (int j = 0; j < 4; ++j) patch[(4 * h + j) * TW + px] = v[j];
                }
            }
            // read the 8 x TW block back flat and store 16-byte pieces of its rows (one contiguous run when TW == P).
            // Exactly NRB store instructions per register quad, whatever the tile: wait_w counts on it.
            float *out_g = out_s + (int64_t)(8 * gq) * P;
#pragma unroll
            for (int k = 0; k < NRB; ++k) {
                const float4 o4 = *reinterpret_cast<const float4 *>(patch + 4 * (64 * k + lane));
                if (64 * k + 64 <= 8 * G::PPR || 64 * k + lane < 8 * G::PPR) *reinterpret_cast<float4 *>(out_g + rb_off[k]) = o4;
            }
        }
    };

    // strip s + 1's weights are requested, strip s is stored, strip s + 1 is multiplied.  The wait for the weights leaves
    // the strip's 4 NRB stores in flight (vmcnt(4 NRB)): no store acknowledgement is ever waited for.
    // S_x (per-pixel activation sums) is only needed by strips with some zw' != 0 (asymmetric weights): decided per strip
    // from the zero points that arrived with its weights -- no pass over w_zero in the prologue (a serial chain of
    // dependent loads in front of the first barrier)
    bool sx_cur = __builtin_amdgcn_ballot_w64((c_zw - zw_shift) != 0.0f) != 0ull;
    if (sx_cur) mma_strip(std::true_type{}); else mma_strip(std::false_type{});
    QE_ST(3);   // K loops
    for (int s = 0; s + 1 < n_my; ++s) {
        const float e_sw = c_sw, e_zw = c_zw, e_bi = c_bi;
        load_w(strip0 + (s + 1) * WAVES);
        epilogue(strip0 + s * WAVES, e_sw, e_zw, e_bi, sx_cur);
        QE_ST(4);   // epilogues: conversions, patch round trips, stores issued
        wait_w(std::integral_constant<int, N_YOUNGER>{});
        QE_ST(5);   // wait for the next strip's weights
        sx_cur = __builtin_amdgcn_ballot_w64((c_zw - zw_shift) != 0.0f) != 0ull;
        if (sx_cur) mma_strip(std::true_type{}); else mma_strip(std::false_type{});
        QE_ST(3);
    }
    epilogue(strip0 + (n_my - 1) * WAVES, c_sw, c_zw, c_bi, sx_cur);
    if constexpr (RQ) rq_report(a, bad);
#ifdef QE_STAMP
    QE_ST(4);
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): stores acknowledged
    QE_ST(6);   // store drain
    if (a.dbg != nullptr && lane == 0) {
        unsigned long long *o = a.dbg + ((size_t)blockIdx.x * WAVES + wave) * 10;
        for (int i = 0; i < 8; ++i) o[i] = st[i];
        o[8] = tprev - tstart;
        o[9] = tstart;
    }
#endif
}

// (A persistent form -- one 8-wave workgroup per CU walking the tiles, the next tile requested by LDS-DMA one or two tiles ahead
// into a second / third buffer -- was built, held to the parity bar and measured 10-23 % SLOWER than one tile per workgroup
// (profiles/r03d_ab_persist.txt, r03t_ab_persist2.txt): the second resident workgroup of the plain form already overlaps more
// than the prefetch buys.  Removed; git history at eb1114b.)

// ---------------------------------------------------------------------------------------------
// 7x7 planes (P = 49: the last stage of a bottleneck network, 512 -> 2048 @7x7).  Same scheme on the small-plane layout of
// qe_conv_flatd.hip: a tile is GI whole images; image gi, channel c owns a 64-byte LDS row whose four 16-byte slots sit at
// position j ^ 2 ((c >> 2) & 1), so the 8 rows x 2 pixel groups of a transposed read cover all 64 banks once; a plane is 49
// contiguous bytes at byte alignment, fetched by LDS-DMA as bytes [16 j, 16 j + 16) (the slack of slot 3 is the next plane's
// bytes, never stored; only the tensor's last byte has to be patched in by hand).  Column tiles 2 gi, 2 gi + 1 are pixels
// 0-31 / 32-48 of image gi.  A strip's 32 channels x 49 pixels of one image are ONE contiguous, line-aligned 6,272-byte run
// of the output: the wave lays the run out in its LDS patch exactly as it stands in memory and copies it flat.
// ---------------------------------------------------------------------------------------------
template <int KS, int GI> struct Pwr7Geom {
    static constexpr int WAVES = 8;
    static constexpr int IC = 32 * KS;
    static constexpr int IMG = IC * 64;                      // LDS bytes of one image
    static constexpr int XBYTES = GI * IMG;
    static constexpr int XINSTR = XBYTES / 1024;
    static constexpr int PXW = (XINSTR + WAVES - 1) / WAVES;
    static constexpr int NT = 2 * GI;
    static constexpr int TAB = WAVES * 4 * 32 * 4;
    static constexpr int PATCH = 32 * 49 * 4;
    static constexpr int NRB = 7;                            // 392 16-byte pieces of a run in rounds of 64 lanes
    static constexpr int LDS = XBYTES + TAB + WAVES * PATCH;
};

// RQ: fused re-quantisation (qe_quantconv2d_requant_prepared): the strip's 32 planes x 49 codes of an image are ONE contiguous,
// 16-byte aligned 1,568-byte run of the output; the wave lays the codes out in its patch as they stand in memory (byte writes)
// and copies the run flat: 2 store instructions per image and strip instead of 7.
template <int KS, int GI, bool RQ = false>
__global__ __launch_bounds__(512, 2) void conv_pwr7_kernel(const PwrArgs a)
{
    using G = Pwr7Geom<KS, GI>;
    constexpr int WAVES = 8, NT = G::NT, PXW = G::PXW, NRB = G::NRB, P = 49;
    static_assert(G::XINSTR % WAVES == 0, "every wave issues the same number of tile pieces");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, h = lane >> 5;

    int pt, og;
    {
        const int bid = blockIdx.x;
        const int idx = bid >> 3;
        const int j = idx / a.n_groups;
        og = idx - j * a.n_groups;
        const int c = j / a.chunk;
        pt = (c * 8 + (bid & 7)) * a.chunk + (j - c * a.chunk);
    }
    if (pt >= a.n_pix_tiles) return;
    const int n0 = pt * GI;                                   // host: N % GI == 0

    uint8_t *Xs = smem;
    float *tab = reinterpret_cast<float *>(smem + G::XBYTES) + wave * 128;
    float *patch = reinterpret_cast<float *>(smem + G::XBYTES + G::TAB) + wave * (32 * P);

    const int strip0 = og * a.strips_per_group + wave;
    const int n_my = (a.strips_per_group - wave + WAVES - 1) / WAVES;
    v4i wf[KS];
    float c_sw, c_zw, c_bi;
    auto load_w = [&](int strip) __attribute__((always_inline)) {
        const int oc = strip * 32 + col;
        const float *psw = a.w_scale + (a.w_per_tensor ? 0 : oc);
        const float *pzw = a.w_zero + (a.w_per_tensor ? 0 : oc);
        const float *pbi = a.bias ? a.bias + oc : psw;
        const uint8_t *wl = a.w + (int64_t)oc * G::IC + 16 * h;
        asm volatile("global_load_dword %0, %3, off\n\tglobal_load_dword %1, %4, off\n\tglobal_load_dword %2, %5, off"
                     : "=&v"(c_sw), "=&v"(c_zw), "=&v"(c_bi) : "v"(psw), "v"(pzw), "v"(pbi) : "memory");
#define QE_PWR_LW(K, OFF) if constexpr (KS > K) asm volatile("global_load_dwordx4 %0, %1, off offset:" #OFF : "=v"(wf[K]) : "v"(wl) : "memory")
        QE_PWR_LW(0, 0); QE_PWR_LW(1, 32); QE_PWR_LW(2, 64); QE_PWR_LW(3, 96);
        QE_PWR_LW(4, 128); QE_PWR_LW(5, 160); QE_PWR_LW(6, 192); QE_PWR_LW(7, 224);
        QE_PWR_LW(8, 256); QE_PWR_LW(9, 288); QE_PWR_LW(10, 320); QE_PWR_LW(11, 352);
        QE_PWR_LW(12, 384); QE_PWR_LW(13, 416); QE_PWR_LW(14, 448); QE_PWR_LW(15, 480);
#undef QE_PWR_LW
    };
    // one wait statement naming every destination (section 5.7 form (ii)); N = stores issued behind the requests
#define QE_PWR7_WAIT(N)                                                                                                   \
    do {                                                                                                                  \
        if constexpr (KS == 4) asm volatile("s_waitcnt vmcnt(%7)" : "+v"(c_sw), "+v"(c_zw), "+v"(c_bi), "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]) : "i"(N) : "memory"); \
        else if constexpr (KS == 8) asm volatile("s_waitcnt vmcnt(%11)" : "+v"(c_sw), "+v"(c_zw), "+v"(c_bi), "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]), "+v"(wf[KS - 4]), "+v"(wf[KS - 3]), "+v"(wf[KS - 2]), "+v"(wf[KS - 1]) : "i"(N) : "memory"); \
        else asm volatile("s_waitcnt vmcnt(%19)" : "+v"(c_sw), "+v"(c_zw), "+v"(c_bi), "+v"(wf[0]), "+v"(wf[1]), "+v"(wf[2]), "+v"(wf[3]), "+v"(wf[4]), "+v"(wf[5]), "+v"(wf[6]), "+v"(wf[7]), \
                          "+v"(wf[KS - 8]), "+v"(wf[KS - 7]), "+v"(wf[KS - 6]), "+v"(wf[KS - 5]), "+v"(wf[KS - 4]), "+v"(wf[KS - 3]), "+v"(wf[KS - 2]), "+v"(wf[KS - 1]) : "i"(N) : "memory"); \
        if (!a.bias) c_bi = 0.0f;                                                                                        \
    } while (0)
    if (n_my > 0) load_w(strip0);

    // ---- the tile by LDS-DMA: slot e = 64 * (wave + 8 i) + lane <-> (image, channel c, position jj) ----
    const int64_t x_total = (int64_t)a.N * G::IC * P;
    const int64_t x_last16 = x_total - 16;
    const uint32_t tail_byte = a.x[x_total - 1];
    int fix_i = -1;
#pragma unroll
    for (int i = 0; i < PXW; ++i) {
        const int e = 64 * (wave + WAVES * i) + lane;
        const int img = e / (G::IC * 4);
        const int r = e - img * (G::IC * 4);
        const int c = r >> 2, jj = r & 3;
        const int j = jj ^ (2 * ((c >> 2) & 1));
        const int64_t src = ((int64_t)(n0 + img) * G::IC + c) * P + 16 * j;
        const bool skip = src > x_last16;                     // the tensor's last plane, slot 3: one valid byte (the very last)
        if (skip) fix_i = i;
        if (!skip)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.x + src),
                                             (__attribute__((address_space(3))) void *)(Xs + 1024 * (wave + WAVES * i)), 16, 0, 0);
    }

    const float zxp = a.x_zero[0] - (a.x_sign ? 0.0f : 128.0f);
    const float sx = a.x_scale[0];
    const float zw_shift = a.w_sign ? 0.0f : 128.0f;

    if (n_my > 0) QE_PWR7_WAIT(0); else __builtin_amdgcn_s_waitcnt(0x0f70);
#pragma unroll
    for (int i = 0; i < PXW; ++i) {
        uint4 *slot = reinterpret_cast<uint4 *>(Xs + 1024 * (wave + WAVES * i) + 16 * lane);
        uint4 v = *slot;
        if (i == fix_i) v = make_uint4(tail_byte, 0u, 0u, 0u);
        v.x ^= 0x80808080u; v.y ^= 0x80808080u; v.z ^= 0x80808080u; v.w ^= 0x80808080u;
        *slot = v;
    }
    __syncthreads();
    if (n_my <= 0) return;

    // transposed-read bases (qe_conv_flatd.hip, SMALL): row (16 h + i16 / 2) of a 32-channel step; logical slot 2 tt + ((lane >> 4) & 1)
    // sits at position slot ^ 2 for rows 4-7 of every 8
    const int i16 = lane & 15;
    const int sm_sl = ((lane >> 4) & 1) ^ (2 * (i16 >> 3));
    const int rowb = (16 * h + (i16 >> 1)) * 64 + 8 * (i16 & 1);
    const uint8_t *tr_b0 = Xs + rowb + 16 * sm_sl;
    const uint8_t *tr_b1 = Xs + rowb + 16 * (sm_sl ^ 2);

    float sxv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) sxv[t] = 0.0f;
    v16i acc[NT];
    int swacc;
    auto mma_strip = [&](auto sx_tag) __attribute__((always_inline)) {
        constexpr bool SX = decltype(sx_tag)::value;
        int sxacc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            sxacc[t] = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0;
        }
        swacc = 0;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            v4i wk = wf[k];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                wk[j] ^= (int)0x80808080;
                swacc = __builtin_amdgcn_sdot4(wk[j], 0x01010101, swacc, false);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const uint8_t *src = ((t & 1) ? tr_b1 : tr_b0) + (t >> 1) * G::IMG + k * (32 * 64);
                const v2i lo = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3))) *)(src));
                const v2i hi = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3))) *)(src + 8 * 64));
                const v4i xf = {lo[0], lo[1], hi[0], hi[1]};
                if constexpr (SX) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) sxacc[t] = __builtin_amdgcn_sdot4(xf[j], 0x01010101, sxacc[t], false);
                }
                acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wk, xf, acc[t], 0, 0, 0);
            }
        }
        if constexpr (SX) {
#pragma unroll
            for (int t = 0; t < NT; ++t) sxv[t] = (float)(sxacc[t] + __shfl_xor(sxacc[t], 32));
        }
    };

    RqConst rqc;
    bool bad = false, rq_fast_u = false;
    if constexpr (RQ) {
        rqc = rq_setup(a);
        rqc.slow = __builtin_amdgcn_readfirstlane(rqc.slow);
        rqc.chk = __builtin_amdgcn_readfirstlane(rqc.chk);
        rq_fast_u = rq_fast_ok(rqc);
    }
    auto epilogue = [&](int strip, float e_sw, float e_zw, float e_bi, bool need_sx) __attribute__((always_inline)) {
        const int oc0 = strip * 32;
        bool fast = false;
        {
            const float zwp = e_zw - zw_shift;
            const int sw_sum = swacc + __shfl_xor(swacc, 32);
            const float cst = fmaf((float)G::IC * zxp, zwp, -zxp * (float)sw_sum);
            if (h == 0) {
                tab[col] = sx * e_sw;
                tab[32 + col] = cst;
                tab[64 + col] = e_bi;
                tab[96 + col] = zwp;
            }
            if constexpr (RQ) fast = rq_fast_u && __builtin_amdgcn_ballot_w64(!rq_bounded(sx * e_sw, cst, e_bi, zwp)) == 0ull;
        }
        if constexpr (RQ) {
            uint8_t *bp = reinterpret_cast<uint8_t *>(patch);                  // [32][49] codes of one image, as in memory
#pragma unroll
            for (int gi = 0; gi < GI; ++gi) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 al = *reinterpret_cast<const float4 *>(tab + 8 * gq + 4 * h);
                    const float4 cs = *reinterpret_cast<const float4 *>(tab + 32 + 8 * gq + 4 * h);
                    const float4 bi = *reinterpret_cast<const float4 *>(tab + 64 + 8 * gq + 4 * h);
                    const float alv[4] = {al.x, al.y, al.z, al.w}, csv[4] = {cs.x, cs.y, cs.z, cs.w};
                    const float biv[4] = {bi.x, bi.y, bi.z, bi.w};
                    float zwv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (need_sx) {
                        const float4 zw = *reinterpret_cast<const float4 *>(tab + 96 + 8 * gq + 4 * h);
                        zwv[0] = zw.x; zwv[1] = zw.y; zwv[2] = zw.z; zwv[3] = zw.w;
                    }
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        const int t = 2 * gi + tt;
                        const int px = 32 * tt + col;
                        const bool real = tt == 0 || px < P;
                        float y[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            // the fp32 form's operation order, then quantised as quantize_pack would
                            float f = (float)acc[t][4 * gq + j] + csv[j];
                            if (need_sx) f = fmaf(-zwv[j], sxv[t], f);
                            y[j] = fmaf(alv[j], f, biv[j]);
                        }
                        if (fast) {
#pragma unroll
                            for (int j = 0; j < 4; j += 2) {
                                const v2f c2 = rq_fast2(rqc, v2f{y[j], y[j + 1]});
                                if (real) {
                                    bp[(8 * gq + 4 * h + j) * P + px] = (uint8_t)(unsigned)c2.x;
                                    bp[(8 * gq + 4 * h + j + 1) * P + px] = (uint8_t)(unsigned)c2.y;
                                }
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                bool b = false;
                                const float r = rq_value(rqc, y[j], b);
                                bad |= b && real;
                                if (real) bp[(8 * gq + 4 * h + j) * P + px] = (uint8_t)(unsigned)r;
                            }
                        }
                    }
                }
                // the run of image n0 + gi: 32 x 49 codes = 98 16-byte pieces, copied flat
                uint8_t *dst = a.rq_out + ((int64_t)(n0 + gi) * a.OC + oc0) * P;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const uint4 d4 = *reinterpret_cast<const uint4 *>(bp + 16 * (64 * k + lane));
                    if (64 * k + lane < 98) *reinterpret_cast<uint4 *>(dst + 16 * (64 * k + lane)) = d4;
                }
            }
            return;
        }
#pragma unroll
        for (int gi = 0; gi < GI; ++gi) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 al = *reinterpret_cast<const float4 *>(tab + 8 * gq + 4 * h);
                const float4 cs = *reinterpret_cast<const float4 *>(tab + 32 + 8 * gq + 4 * h);
                const float4 bi = *reinterpret_cast<const float4 *>(tab + 64 + 8 * gq + 4 * h);
                const float alv[4] = {al.x, al.y, al.z, al.w}, csv[4] = {cs.x, cs.y, cs.z, cs.w};
                const float biv[4] = {bi.x, bi.y, bi.z, bi.w};
                float zwv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                if (need_sx) {
                    const float4 zw = *reinterpret_cast<const float4 *>(tab + 96 + 8 * gq + 4 * h);
                    zwv[0] = zw.x; zwv[1] = zw.y; zwv[2] = zw.z; zwv[3] = zw.w;
                }
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int t = 2 * gi + tt;
                    const int px = 32 * tt + col;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // the flat kernels' operation order (bit-identity with their fused re-quantising epilogue)
                        float f = (float)acc[t][4 * gq + j] + csv[j];
                        if (need_sx) f = fmaf(-zwv[j], sxv[t], f);
                        const float v = fmaf(alv[j], f, biv[j]);
                        if (tt == 0 || px < P) patch[(8 * gq + 4 * h + j) * P + px] = v;
                    }
                }
            }
            // the run of image n0 + gi: 32 rows x 49 floats = 392 16-byte pieces, copied flat (NRB stores, the last one 8 lanes wide)
            float *dst = a.out + ((int64_t)(n0 + gi) * a.OC + oc0) * P;
#pragma unroll
            for (int k = 0; k < NRB; ++k) {
                const float4 o4 = *reinterpret_cast<const float4 *>(patch + 4 * (64 * k + lane));
                if (64 * k + 64 <= 392 || 64 * k + lane < 392) *reinterpret_cast<float4 *>(dst + 4 * (64 * k + lane)) = o4;
            }
        }
    };

    bool sx_cur = __builtin_amdgcn_ballot_w64((c_zw - zw_shift) != 0.0f) != 0ull;
    if (sx_cur) mma_strip(std::true_type{}); else mma_strip(std::false_type{});
    for (int s = 0; s + 1 < n_my; ++s) {
        const float e_sw = c_sw, e_zw = c_zw, e_bi = c_bi;
        load_w(strip0 + (s + 1) * WAVES);
        epilogue(strip0 + s * WAVES, e_sw, e_zw, e_bi, sx_cur);
        QE_PWR7_WAIT((RQ ? 2 : NRB) * GI);                    // the strip's stores stay in flight
        sx_cur = __builtin_amdgcn_ballot_w64((c_zw - zw_shift) != 0.0f) != 0ull;
        if (sx_cur) mma_strip(std::true_type{}); else mma_strip(std::false_type{});
    }
    epilogue(strip0 + (n_my - 1) * WAVES, c_sw, c_zw, c_bi, sx_cur);
    if constexpr (RQ) rq_report(a, bad);
#undef QE_PWR7_WAIT
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
extern unsigned long long *g_mfma_dbg;   // qe_conv_mfma.hip (diagnostic builds)

struct PwrPlan {
    int tw = 0, waves = 0, ks = 0, groups = 1;
    bool s2 = false;
};

// QE_PWR=0 disables the kernel, QE_PWR_GROUPS overrides the channel split (tuning).
static bool pwr_plan(const qe_conv_shape *sh, const qe_qparam *x, const qe_qparam *w, PwrPlan *pl)
{
    // QE_PWR=0: never; QE_PWR=1: only layers whose planes are ONE tile (14x14: the tile is fetched once instead of OC/128
    // times and every strip leaves as one contiguous run: -20..25 % against the flat kernels, profiles/r03a_ab_pwr.txt) and
    // the stride-2 layers (the strided rows fetched once per tile instead of once per 128 output channels); default (2):
    // every eligible layer -- on stride-1 28x28 / 56x56 planes both kernels sit near the same store rate layer by layer
    // (+-3 %, inside the noise of isolated timings), over the whole step this form is 1.2 % ahead (three alternating pairs of
    // 200-step runs on one box, profiles/r03t_ab_pwr_stack.txt)
    int mode = 2;
    if (const char *e = env_get("QE_PWR")) mode = atoi(e);
    if (mode == 0) return false;
    if (sh->KH != 1 || sh->KW != 1 || sh->padding != 0 || (sh->stride != 1 && sh->stride != 2)) return false;
    if (x->n_bits != 8 || w->n_bits != 8 || x->n_param != 1) return false;
    if (sh->IC != 64 && sh->IC != 128 && sh->IC != 256) return false;
    if (sh->OC % 32 != 0 || sh->OC < 128 || sh->N < 1) return false;
    const bool s2 = sh->stride == 2;
    if (s2 && ((sh->H & 1) || (sh->W & 7) || sh->W > 64 || (env_get("QE_PWR_S2") && atoi(env_get("QE_PWR_S2")) == 0))) return false;
    const int OH = s2 ? sh->H / 2 : sh->H, OW = s2 ? sh->W / 2 : sh->W;
    const int64_t P = (int64_t)OH * OW;                       // output plane
    // tiles of 224 or 196 pixels that divide the plane (56x56: 14 x 224; 28x28: 4 x 196; 14x14: the plane itself)
    const int tw = (P % 224 == 0) ? 224 : ((P % 196 == 0) ? 196 : 0);
    if (tw == 0) return false;
    if (s2 && (tw % OW != 0 || tw / OW > 8)) return false;    // whole output rows per tile
    if (mode == 1 && !s2 && P != tw) return false;
    if ((int64_t)sh->N * sh->IC * sh->H * sh->W < 16 || (int64_t)sh->OC * P >= (1ll << 29) || (int64_t)sh->IC * sh->H * sh->W >= (1ll << 31)) return false;
    if ((reinterpret_cast<uintptr_t>(w->data) & 15) != 0 || (reinterpret_cast<uintptr_t>(x->data) & (s2 ? 7 : 3)) != 0) return false;
    if ((reinterpret_cast<uintptr_t>(w->scale) & 3) != 0) return false;
    const int ks = sh->IC / 32;
    const int waves = ks == 8 ? 8 : 4;                        // IC = 256: 56 KB of tile -> one 8-wave workgroup per CU
    if (sh->OC < 64 * waves) return false;                    // fewer than two strips per wave: the flat kernels' tiling fits better (256 -> 128 @56x56: +19 %)
    int groups = 1;
    const int strips = sh->OC / 32;
    if (const char *e = env_get("QE_PWR_GROUPS")) { const int v = atoi(e); if (v >= 1 && strips % v == 0) groups = v; }
    pl->tw = tw; pl->waves = waves; pl->ks = ks; pl->groups = groups; pl->s2 = s2;
    return true;
}

// 7x7 planes: 0 = not eligible, else images per tile
static int pwr7_plan(const qe_conv_shape *sh, const qe_qparam *x, const qe_qparam *w, int *groups)
{
    int mode = 2;
    if (const char *e = env_get("QE_PWR")) mode = atoi(e);
    if (mode == 0 || (env_get("QE_PWR7") && atoi(env_get("QE_PWR7")) == 0)) return 0;
    if (sh->KH != 1 || sh->KW != 1 || sh->stride != 1 || sh->padding != 0 || sh->H * sh->W != 49) return 0;
    if (x->n_bits != 8 || w->n_bits != 8 || x->n_param != 1) return 0;
    if (sh->IC != 128 && sh->IC != 256 && sh->IC != 512) return 0;
    const int gi = sh->IC == 512 ? 2 : 4;                     // 64 KB of tile
    if (sh->OC % 32 != 0 || sh->OC < 512 || sh->N < gi || sh->N % gi != 0) return 0;   // >= 2 strips per wave; whole tiles only
    if ((int64_t)sh->N * sh->IC * 49 >= (1ll << 31) || (int64_t)sh->OC * 49 >= (1ll << 29)) return 0;
    if ((reinterpret_cast<uintptr_t>(w->data) & 15) != 0 || (reinterpret_cast<uintptr_t>(x->data) & 15) != 0) return 0;
    const int tiles = sh->N / gi, strips = sh->OC / 32;
    int g = 1;
    while (tiles * g < kNumCU && strips % (2 * g) == 0 && strips / (2 * g) >= 8) g *= 2;   // about one workgroup per CU, >= 1 strip per wave
    if (const char *e = env_get("QE_PWR_GROUPS")) { const int v = atoi(e); if (v >= 1 && strips % v == 0) g = v; }
    *groups = g;
    return gi;
}

// rq != nullptr: the fused re-quantising form (8-bit codes, one scale: what the kernels' epilogue covers)
bool pwr_eligible(const qe_conv_shape *sh, const qe_qparam *x, const qe_qparam *w, const RequantHost *rq)
{
    PwrPlan pl;
    int g;
    if (rq != nullptr) {
        if (rq->n_bits != 8 || rq->n_param != 1 || rq->out == nullptr || (reinterpret_cast<uintptr_t>(rq->out) & 15) != 0) return false;
        if (env_get("QE_PWR_RQ") && atoi(env_get("QE_PWR_RQ")) == 0) return false;
        return pwr_plan(sh, x, w, &pl) || pwr7_plan(sh, x, w, &g) != 0;
    }
    return pwr_plan(sh, x, w, &pl) || pwr7_plan(sh, x, w, &g) != 0;
}

static int launch_pwr7(const qe_qparam *x, const qe_qparam *w, const float *bias, const qe_conv_shape *sh, float *out, hipStream_t s,
                       const RequantHost *rq)
{
    int groups = 1;
    const int gi = pwr7_plan(sh, x, w, &groups);
    if (gi == 0 || (rq == nullptr && (reinterpret_cast<uintptr_t>(out) & 15) != 0)) return QE_ERR_UNSUPPORTED;
    PwrArgs a;
    a.rq_out = nullptr; a.rq_scale = nullptr; a.rq_zero = nullptr; a.rq_status = nullptr;
    a.rq_qmin = a.rq_qmax = a.rq_lo = a.rq_hi = 0.0f; a.rq_offset = 0;
    if (rq != nullptr) {
        if (!pwr_eligible(sh, x, w, rq)) return QE_ERR_UNSUPPORTED;
        a.rq_out = rq->out; a.rq_scale = rq->scale; a.rq_zero = rq->zero;
        a.rq_qmin = rq->qmin; a.rq_qmax = rq->qmax; a.rq_status = rq->status;
        a.rq_offset = rq->sign ? 128u : 0u;                       // tpack.cu:108-111
        a.rq_lo = rq->sign ? -128.0f : 0.0f; a.rq_hi = rq->sign ? 127.0f : 255.0f;
    }
    a.W_in = 7; a.PIN = 49; a.OW = 7;
    a.x = static_cast<const uint8_t *>(x->data); a.w = static_cast<const uint8_t *>(w->data);
    a.x_scale = x->scale; a.x_zero = x->zero; a.w_scale = w->scale; a.w_zero = w->zero; a.bias = bias;
    a.x_sign = x->sign; a.w_sign = w->sign; a.w_per_tensor = (w->n_param == 1);
    a.out = out; a.N = sh->N; a.IC = sh->IC; a.OC = sh->OC; a.P = 49;
    a.tiles_per_image = 1;
    a.n_pix_tiles = sh->N / gi;
    a.n_groups = groups;
    a.strips_per_group = sh->OC / 32 / groups;
    a.dbg = g_mfma_dbg;
    const int64_t per_xcd = ((int64_t)a.n_pix_tiles + 7) / 8;
    a.chunk = (int)(per_xcd < 1 ? 1 : per_xcd);
    const int64_t runs = ((int64_t)a.n_pix_tiles + a.chunk - 1) / a.chunk;
    const int64_t blocks = (runs + 7) / 8 * a.chunk * 8 * a.n_groups;
    if (blocks > 0x7fffffffLL) return QE_ERR_UNSUPPORTED;
#define QE_PWR7_LAUNCH2(KSV, GIV, RQV)                                                                                      \
    do {                                                                                                                    \
        static const bool ok_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_pwr7_kernel<KSV, GIV, RQV>),      \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, Pwr7Geom<KSV, GIV>::LDS) == hipSuccess; \
        (void)ok_;                                                                                                          \
        constexpr size_t lds_ = Pwr7Geom<KSV, GIV>::LDS;                                                                    \
        hipLaunchKernelGGL((conv_pwr7_kernel<KSV, GIV, RQV>), dim3((unsigned)blocks), dim3(512), lds_, s, a);               \
    } while (0)
#define QE_PWR7_LAUNCH(KSV, GIV) do { if (rq != nullptr) QE_PWR7_LAUNCH2(KSV, GIV, true); else QE_PWR7_LAUNCH2(KSV, GIV, false); } while (0)
    if (sh->IC == 512) QE_PWR7_LAUNCH(16, 2); else if (sh->IC == 256) QE_PWR7_LAUNCH(8, 4); else QE_PWR7_LAUNCH(4, 4);
#undef QE_PWR7_LAUNCH
#undef QE_PWR7_LAUNCH2
    QE_LAUNCH_CHECK();
    return QE_OK;
}

int launch_pwr(const qe_qparam *x, const qe_qparam *w, const float *bias, const qe_conv_shape *sh, float *out, hipStream_t s,
               const RequantHost *rq)
{
    if (sh->H * sh->W == 49) return launch_pwr7(x, w, bias, sh, out, s, rq);
    PwrPlan pl;
    if (!pwr_plan(sh, x, w, &pl)) return QE_ERR_UNSUPPORTED;
    PwrArgs a;
    a.rq_out = nullptr; a.rq_scale = nullptr; a.rq_zero = nullptr; a.rq_status = nullptr;
    a.rq_qmin = a.rq_qmax = a.rq_lo = a.rq_hi = 0.0f; a.rq_offset = 0;
    if (rq != nullptr) {
        if (!pwr_eligible(sh, x, w, rq)) return QE_ERR_UNSUPPORTED;
        a.rq_out = rq->out; a.rq_scale = rq->scale; a.rq_zero = rq->zero;
        a.rq_qmin = rq->qmin; a.rq_qmax = rq->qmax; a.rq_status = rq->status;
        a.rq_offset = rq->sign ? 128u : 0u;                       // tpack.cu:108-111
        a.rq_lo = rq->sign ? -128.0f : 0.0f; a.rq_hi = rq->sign ? 127.0f : 255.0f;
    }
    a.x = static_cast<const uint8_t *>(x->data); a.w = static_cast<const uint8_t *>(w->data);
    a.x_scale = x->scale; a.x_zero = x->zero; a.w_scale = w->scale; a.w_zero = w->zero; a.bias = bias;
    a.x_sign = x->sign; a.w_sign = w->sign; a.w_per_tensor = (w->n_param == 1);
    a.out = out; a.N = sh->N; a.IC = sh->IC; a.OC = sh->OC;
    a.W_in = sh->W; a.PIN = sh->H * sh->W;
    a.OW = pl.s2 ? sh->W / 2 : sh->W;
    a.P = pl.s2 ? (sh->H / 2) * (sh->W / 2) : sh->H * sh->W;
    a.tiles_per_image = a.P / pl.tw;
    a.n_pix_tiles = sh->N * a.tiles_per_image;
    a.n_groups = pl.groups;
    a.strips_per_group = sh->OC / 32 / pl.groups;
    a.dbg = g_mfma_dbg;
    const int64_t per_xcd = ((int64_t)a.n_pix_tiles + 7) / 8;
    a.chunk = (int)(per_xcd < 1 ? 1 : per_xcd);
    if (const char *ci = env_get("QE_CHUNK_IMAGES")) {
        const int64_t k = (int64_t)atoi(ci) * a.tiles_per_image;
        a.chunk = (int)(k < 1 ? 1 : (k < per_xcd ? k : per_xcd));
    }
    const int64_t runs = ((int64_t)a.n_pix_tiles + a.chunk - 1) / a.chunk;
    const int64_t blocks = (runs + 7) / 8 * a.chunk * 8 * a.n_groups;
    if (blocks > 0x7fffffffLL) return QE_ERR_UNSUPPORTED;
#define QE_PWR_LAUNCH2(WV, KSV, TWV, S2V, RQV)                                                                             \
    do {                                                                                                                    \
        static const bool ok_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_pwr_kernel<7, WV, KSV, TWV, S2V, RQV>), \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, PwrGeom<7, WV, KSV, TWV>::LDS) == hipSuccess; \
        (void)ok_;                                                                                                          \
        constexpr size_t lds_ = PwrGeom<7, WV, KSV, TWV>::LDS;                                                              \
        hipLaunchKernelGGL((conv_pwr_kernel<7, WV, KSV, TWV, S2V, RQV>), dim3((unsigned)blocks), dim3(64 * WV), lds_, s, a); \
    } while (0)
#define QE_PWR_LAUNCH1(WV, KSV, TWV, S2V) do { if (rq != nullptr) QE_PWR_LAUNCH2(WV, KSV, TWV, S2V, true); else QE_PWR_LAUNCH2(WV, KSV, TWV, S2V, false); } while (0)
#define QE_PWR_LAUNCH(WV, KSV, TWV) do { if (pl.s2) QE_PWR_LAUNCH1(WV, KSV, TWV, true); else QE_PWR_LAUNCH1(WV, KSV, TWV, false); } while (0)
    if (pl.tw == 224) {
        if (pl.ks == 2) QE_PWR_LAUNCH(4, 2, 224); else if (pl.ks == 4) QE_PWR_LAUNCH(4, 4, 224); else QE_PWR_LAUNCH(8, 8, 224);
    } else {
        if (pl.ks == 2) QE_PWR_LAUNCH(4, 2, 196); else if (pl.ks == 4) QE_PWR_LAUNCH(4, 4, 196); else QE_PWR_LAUNCH(8, 8, 196);
    }
#undef QE_PWR_LAUNCH1
#undef QE_PWR_LAUNCH2
#undef QE_PWR_LAUNCH
    QE_LAUNCH_CHECK();
    return QE_OK;
}

}  // namespace qe
