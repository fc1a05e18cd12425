// qe_conv_c3.hip -- 3x3 / stride 1 / pad 1 quantconv2d with the WHOLE halo image (every input channel) resident in LDS.
//
// Replaces the per-element loop of quantconv2d_cuda_kernel (engine/kernels/functions/quantconv2d.cu:78-141) for the
// MFMA-bound 3x3 layers of a bottleneck network's 14x14 stage (256 -> 256: 5 of the 53 ResNet-50 convolutions).  Same
// integer reformulation, operand layout and epilogue as conv_mfma_sm2_kernel (qe_conv_mfma_kernel.hpp):
//   out = bias + sx sw ( S_aw - zw' S_x - zx' S_w + N_inb zx' zw' ),  S_aw exact in int32 on v_mfma_i32_32x32x32_i8,
//   A = weights (rows = output channel, fragment order Wt[tap][ic/16][oc][16] of the prep pass),
//   B = activations from a pixel-major halo image ([16-channel group][pixel][16 B], zero borders: a tap is an LDS offset).
//
// Why another kernel.  In-kernel stamps of the sm2 kernel on 256 -> 256 @14x14 (profiles/r03i_stamps.txt): a wave spends
// 36 % of its life in the MFMA phases; the rest is the per-stage machinery of a K loop that is staged 32 channels at a
// time -- waiting for the stage's activations and weights (23 %), issuing the next stage's loads against the CU's busy
// memory pipe (13 %), two barriers per stage (7 %) -- and the epilogue (17 %).  The image of one 14x14 plane is small:
// ALL 256 channels of it are 64 KB of LDS.  So here
//   * a workgroup (8 waves) owns one image x 256 output channels; the image is fetched, transposed to pixel-major order and
//     written to LDS ONCE, every load of it requested before the first is waited for; one barrier;
//   * after that barrier the waves never synchronise again until the epilogue's tables: wave w owns the 32-channel strip w
//     and runs the whole K loop (9 taps x 8 channel steps x 7 column tiles = 504 MFMAs) on its own.  Its weight
//     fragments come straight from L2 to VGPRs, the nine taps of channel step k + 1 requested (inline asm, counted by
//     hand: section 5.7 of the guide) before the 63 MFMAs of step k -- a 2,000-cycle cover for an L2 round trip;
//   * activations cross the CU's memory path once per image, weights once per (image, strip).
#include "qe_conv_mfma_kernel.hpp"

#include <cstdlib>

namespace qe {

constexpr int C3_THREADS = 512;
constexpr int C3_WAVES = 8;

// nine weight fragments of one 32-channel step: SGPR base (wave-uniform: tap, step) + ONE per-lane 32-bit offset
#define QE_C3_LOADW(W, BASE)                                                                                              \
    do {                                                                                                                  \
        const int8_t *b_ = (BASE);                                                                                        \
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(W[0]) : "v"(w_voff), "s"(b_) : "memory");                  \
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(W[1]) : "v"(w_voff), "s"(b_ + tap_stride) : "memory");     \
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(W[2]) : "v"(w_voff), "s"(b_ + 2 * tap_stride) : "memory"); \
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(W[3]) : "v"(w_voff), "s"(b_ + 3 * tap_stride) : "memory"); \
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(W[4]) : "v"(w_voff), "s"(b_ + 4 * tap_stride) : "memory"); \
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(W[5]) : "v"(w_voff), "s"(b_ + 5 * tap_stride) : "memory"); \
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(W[6]) : "v"(w_voff), "s"(b_ + 6 * tap_stride) : "memory"); \
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(W[7]) : "v"(w_voff), "s"(b_ + 7 * tap_stride) : "memory"); \
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(W[8]) : "v"(w_voff), "s"(b_ + 8 * tap_stride) : "memory"); \
    } while (0)
// the K loop holds no other vector-memory operation: vmcnt(0) is exactly "these nine have landed"
#define QE_C3_WAITW(W)                                                                                                    \
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(W[0]), "+v"(W[1]), "+v"(W[2]), "+v"(W[3]), "+v"(W[4]), "+v"(W[5]), "+v"(W[6]), "+v"(W[7]), "+v"(W[8]) : : "memory")

template <int KS, int NIW>
__global__ __launch_bounds__(C3_THREADS, 2) void conv_c3_kernel(const MfmaArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint4 *Xs = reinterpret_cast<uint4 *>(smem);
    constexpr int NG = 2 * KS;                                // 16-channel groups
    constexpr int MT = 32 * C3_WAVES;                         // 256 output channels per workgroup
    constexpr int KK = 9;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, h = lane >> 5;

#ifdef QE_STAMP
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = qe_stamp();
    const unsigned long long tstart = tprev;
#endif
    int pt, ot, th;
    TileGeom g;
    if (!decode_tile(a, pt, ot, g, th)) return;               // GI == 1, TH == OH: one whole image per tile
    const int ISZ = a.IHT * a.IWP;                            // halo pixels of the image
    int *sxp = reinterpret_cast<int *>(Xs + NG * ISZ);        // [ISZ] per-pixel channel sums (asymmetric weights only)

    // ---- weights: this lane's part of every fragment address ------------------------------------------------------
    const int oc_lane = ot * MT + wave * 32 + col;
    const uint32_t w_voff = (uint32_t)(h * a.OCP + oc_lane) * 16u;
    const int64_t grp_stride = (int64_t)a.OCP * 16;
    const int64_t tap_stride = (int64_t)a.NG * grp_stride;
    v4i wa[9], wb[9];
    QE_C3_LOADW(wa, a.wt);                                    // step 0: in flight under the whole staging

    // ---- zero the halo image (borders stay zero: a padded tap multiplies a_x = 0) ---------------------------------
    for (int i = tid; i < NG * ISZ; i += C3_THREADS) Xs[i] = make_uint4(0, 0, 0, 0);
    for (int i = tid; i < ISZ; i += C3_THREADS) sxp[i] = 0;
    int zw_local = 0;
    if (tid < MT) zw_local = (a.ep[a.OCP + ot * MT + tid] != 0.0f) ? 1 : 0;
    const bool need_sx = __syncthreads_or(zw_local) != 0;     // also: the zeroes are in place

    QE_ST(0);   // weight requests, zeroing, first barrier
    // ---- staging: unit u = tid + 512 i <-> (16-channel group, image row, 4-pixel quad), all requested up front -----
    const int NQ = (a.W + 3) >> 2;
    const int HW = a.H * a.W;
    const int UPG = a.H * NQ;                                 // units per channel group
    const int NU = NG * UPG;
    const uint8_t *xi = a.x + (int64_t)g.n0 * a.IC * HW;
    constexpr int UPT = 2;                                    // host: NU <= 2 * 512
    uint32_t d[UPT][16];
    int u_grp[UPT], u_sh[UPT], u_pix[UPT], u_col[UPT];
    bool u_ok[UPT];
#pragma unroll
    for (int i = 0; i < UPT; ++i) {
        const int u = tid + C3_THREADS * i;
        u_ok[i] = u < NU;
        const int uc = u_ok[i] ? u : 0;
        const int gq = uc / UPG, rr = uc - gq * UPG;
        const int l = rr / NQ, iq = rr - l * NQ;
        int iw0 = 4 * iq;
        u_sh[i] = 0;
        if (iw0 + 4 > a.W) { u_sh[i] = 8 * (iw0 + 4 - a.W); iw0 = a.W - 4; }   // last quad ends at the row's end, shifted back below
        u_grp[i] = gq;
        u_col[i] = 4 * iq;
        u_pix[i] = (l + a.pad) * a.IWP + 4 * iq + a.pad;      // halo position of the quad's first pixel
        const uint8_t *p0 = xi + (int64_t)(gq * 16) * HW + l * a.W + iw0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            uint32_t v;
            __builtin_memcpy(&v, p0 + (int64_t)j * HW, 4);
            d[i][j] = v;
        }
    }
#pragma unroll
    for (int i = 0; i < UPT; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) d[i][j] = (d[i][j] >> u_sh[i]) ^ 0x80808080u;   // u - 128: signed q, or unsigned q - 128
        uint32_t o[4][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
            transpose4x4(d[i][4 * m], d[i][4 * m + 1], d[i][4 * m + 2], d[i][4 * m + 3], o[0][m], o[1][m], o[2][m], o[3][m]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (u_ok[i] && u_col[i] + j < a.W) {
                Xs[u_grp[i] * ISZ + u_pix[i] + j] = make_uint4(o[j][0], o[j][1], o[j][2], o[j][3]);
                if (need_sx) {
                    int sum = 0;
#pragma unroll
                    for (int m = 0; m < 4; ++m) sum = __builtin_amdgcn_sdot4((int)o[j][m], 0x01010101, sum, false);
                    atomicAdd(&sxp[u_pix[i] + j], sum);
                }
            }
        }
    }
    QE_ST(1);   // image fetched, transposed, written
    __syncthreads();                                          // the image is complete; from here on every wave runs alone
    QE_ST(2);   // barrier

    // ---- K loop: this wave's strip, 9 taps x KS steps x NIW column tiles --------------------------------------------
    int pixidx[NIW];
#pragma unroll
    for (int t = 0; t < NIW; ++t) {
        const int q = t * 32 + col;
        const int r = q / a.OW, c = q - r * a.OW;
        pixidx[t] = h * ISZ + ((q < g.NT) ? r * a.IWP + c : 0);
    }
    v16i acc[NIW];
#pragma unroll
    for (int t = 0; t < NIW; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0;
    }
    auto mma_step = [&](const v4i (&w)[9], int ks) __attribute__((always_inline)) {
        const uint4 *Xk = Xs + 2 * ks * ISZ;
#pragma unroll
        for (int tap = 0; tap < KK; ++tap) {
            const int off = (tap / 3) * a.IWP + (tap % 3);
#pragma unroll
            for (int t = 0; t < NIW; ++t) {
                const v4i b = *reinterpret_cast<const v4i *>(&Xk[pixidx[t] + off]);
                acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w[tap], b, acc[t], 0, 0, 0);
            }
        }
    };
    QE_C3_WAITW(wa);
    for (int ks = 0; ks < KS; ks += 2) {
        QE_C3_LOADW(wb, a.wt + (int64_t)(2 * (ks + 1)) * grp_stride);
        mma_step(wa, ks);
        QE_C3_WAITW(wb);
        if (ks + 2 < KS) QE_C3_LOADW(wa, a.wt + (int64_t)(2 * (ks + 2)) * grp_stride);
        mma_step(wb, ks + 1);
        if (ks + 2 < KS) QE_C3_WAITW(wa);
    }

    QE_ST(4);   // K loop
    // ---- epilogue: the shared lane = pixel epilogue of the 3x3 kernels ----------------------------------------------
    int sxs[NIW];
#pragma unroll
    for (int t = 0; t < NIW; ++t) {
        sxs[t] = 0;
        if (need_sx) {
            const int pbase = pixidx[t] - h * ISZ;
            for (int tap = 0; tap < KK; ++tap) sxs[t] += sxp[pbase + (tap / 3) * a.IWP + (tap % 3)];
        }
    }
    const float *ctab = stage_ctab<MT>(a, smem, ot, tid, C3_THREADS);
    const int *ptab = ctab ? nullptr : stage_ptab<MT>(a, smem, ot, tid, C3_THREADS);
    mfma_epilogue<C3_WAVES, 1, NIW, false>(a, acc, sxs, need_sx, g, ot, wave, 0, col, h, KK, ptab, ctab);
#ifdef QE_STAMP
    QE_ST(7);   // epilogue: stores issued
    if (a.dbg != nullptr && lane == 0) {
        unsigned long long *o = a.dbg + ((size_t)blockIdx.x * C3_WAVES + wave) * 10;
        for (int i = 0; i < 8; ++i) o[i] = st[i];
        o[8] = tprev - tstart;
        o[9] = tstart;
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// the problem as the sm2 plan describes it (a: geometry, prepared tables) -> eligible?
// OPT-IN (QE_C3=1).  Measured (profiles/r03p_ab_c3.txt, r03q_stamp_c3.txt): bit-identical to the sm2 kernel and no faster, 52.0
// against 49.6-50.2 us on 256 -> 256 @14x14.  The K loop does what it was built for -- 504 MFMAs per wave in 38 k cycles
// with two waves per SIMD = the matrix pipe 85 % busy, no barrier, no exposed load -- but the layer is ONE round of 256
// workgroups that all start together: 16 k cycles of fetch + transposes with the matrix pipe idle, 38 k of MFMAs with the
// memory system idle, 24 k of stores (51 MB at the HBM write rate) with the matrix pipe idle again.  Nothing inside one
// launch overlaps those three: a wave cannot wait for a weight load without waiting for every older store (vmcnt is
// in-order), so an overlapped epilogue needs dedicated store waves fed through LDS -- which at one image per CU hides at
// most the first of two strips' stores (estimate 66 k cycles).
bool c3_eligible(const MfmaArgs &a)
{
    if (!(env_get("QE_C3") && atoi(env_get("QE_C3")) == 1)) return false;
    if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.x_bits != 8) return false;
    if (a.GI != 1 || a.TH != a.OH || a.OH * a.OW > 224 || a.OH * a.OW <= 192) return false;    // one image = 7 column tiles
    if (a.IC != 256 || a.NG != 16 || a.OCP % 256 != 0) return false;
    if (a.W < 4 || a.H * ((a.W + 3) / 4) * a.NG > 2 * C3_THREADS) return false;
    if (a.rq_out != nullptr) return false;
    return true;
}

int launch_c3(const MfmaArgs &a_in, int64_t n_units, hipStream_t s)
{
    MfmaArgs a = a_in;
    a.n_oc_tiles = a.OCP / 256;
    const int ISZ = a.IHT * a.IWP;
    size_t lds = (size_t)a.NG * ISZ * 16 + (size_t)ISZ * 4;
    lds = (lds + 15) / 16 * 16;
    a.ptab_off = (int)lds;                                   // S_w tables of the epilogue (asymmetric activations)
    lds += (size_t)256 * (a.KH + 1) * (a.KW + 1) * sizeof(int);
    a.ctab = (a.n_top + a.n_bot < a.OH && a.n_lft + a.n_rgt < a.OW &&
              (1 + a.n_top + a.n_bot) * (1 + a.n_lft + a.n_rgt) <= (a.KH + 1) * (a.KW + 1) &&
              !(env_get("QE_CTAB") && atoi(env_get("QE_CTAB")) == 0)) ? 1 : 0;
    const int64_t runs = (n_units + a.chunk - 1) / a.chunk;
    const int64_t groups = (runs + 7) / 8 * a.chunk;
    const int64_t blocks = groups * 8 * a.n_oc_tiles;
    constexpr size_t kMaxLds = 96 * 1024;                    // 16 groups x 256 halo pixels x 16 B + sums + tables = 81 KB
    if (blocks > 0x7fffffffLL || lds > kMaxLds) return QE_ERR_UNSUPPORTED;
    static const bool ok_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_c3_kernel<8, 7>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds) == hipSuccess;
    if (!ok_) return QE_ERR_HIP;
    hipLaunchKernelGGL((conv_c3_kernel<8, 7>), dim3((unsigned)blocks), dim3(C3_THREADS), lds, s, a);
    QE_LAUNCH_CHECK();
    return QE_OK;
}

}  // namespace qe
