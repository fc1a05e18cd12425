// qe_conv_halod.hip -- 3x3 / stride 1 / pad 1 quantconv2d with both operands arriving by LDS-DMA ("sm2d").
//
// Replaces the per-element loop of quantconv2d_cuda_kernel (engine/kernels/functions/quantconv2d.cu:78-141) for
// 8-bit activations, 3x3 taps, stride 1, IC % 32 == 0, OC > 64.  Arithmetic, tile geometry, LDS halo image, fragment
// reads, MFMA loop and epilogue are conv_mfma_sm2_kernel's (qe_conv_mfma_kernel.hpp); what differs is the staging:
//
//   sm2 (round 1)                                             sm2d (this file)
//   activations  global -> VGPR dwords (one per channel),     global -> LDS by global_load_lds_dwordx4 in the tensor's own
//                v_perm 4x4 byte transposes, ds_write_b128    [channel][pixel] order (16-byte row pieces, no registers), then
//                = 23 % + 13 % of a wave's life (stamps,      ONE transposing LDS -> LDS pass per stage: ds_read_b64_tr_b8
//                profiles/r02n_stamp_3x3.txt)                 (8 channels x 16 pixels per 16 lanes) + ds_write_b64 into the
//                                                             pixel-major halo image -- 4 + 4 LDS instructions per wave
//   weights      LDS-DMA, ONE buffer (refill waits for the    LDS-DMA, ring of 2 stage buffers: W(s+1) lands while MFMA(s) runs
//                last reader, lands during the next staging)
//   workgroup    4 waves, 128 oc x 1 image, 2 per CU          8 waves, 128 oc x 2 images (14x14) / 14-row band (28x28) /
//                                                             4 images (7x7): half the weight stream per image, 1 per CU;
//                                                             or 4 waves, 2 per CU, single weight buffer (QE_SM2D=2)
//   barriers     2 per 32-channel stage                       1 per stage (8 waves), 2 (4 waves)
//
// MEASURED (profiles/r02o_*): correct on every shape, but NOT faster than sm2 / ws -- 256->256 @14x14 53.4 us (8 waves) and
// 49.8 us (4 waves) against 48.5 us, so it stays opt-in (QE_SM2D=1 | 2, default 0).  What the experiments on it showed:
// removing all DMA saves 5 us, removing the stores 14 us, the MFMAs alone take 22 us and an empty pass 11 us -- the
// phases of a layer that is ONE round of workgroups add up instead of overlapping, whatever the staging costs, and
// delaying half of the workgroups only moves their finish (r02o_sm2d_phase_experiments.txt).
//
// Why a transposing pass instead of transposed fragment reads: ds_read_b64_tr_b8 ignores the low 3 address bits
// (tools/probe_tr_misalign.hip), so a tap's kw = 1, 2 byte shift of a channel-major row cannot be read directly; in the
// pixel-major image a tap is a constant 16-byte-granular offset.  The pass costs 8 LDS instructions per wave and stage.
//
// Stage s (32 channels), every wave:   issue DMA W(s+1) -> Ws[(s+1)&1], X(s+2) -> Xn[s&1];  transpose X(s+1): Xn[(s+1)&1] ->
// Xs[(s+1)&1];  72 MFMAs on Xs[s&1], Ws[s&1];  vmcnt(0) + barrier.  Everything a stage waits for was requested a whole
// MFMA phase earlier.
//
// Natural-order staging of a tile's input rows: channel c of image gi needs the CONTIGUOUS bytes [ih_lo W, ih_hi W) of
// its plane (whole rows), fetched as ceil(L / 16) 16-byte slots, the last one shifted back to END at the last byte (its
// first lanes repeat pixels of the slot before: skipped by the transposing pass) -- nothing is read outside the plane and
// byte-unaligned sources are fine for LDS-DMA (tools/probe_dma_align.hip).
#include "qe_conv_mfma_kernel.hpp"

namespace qe {

typedef int v2i __attribute__((ext_vector_type(2)));

constexpr int SD_MT = 128;
constexpr int SD_WPIECES = 9 * 2 * SD_MT;          // 16-byte weight pieces per stage (36 wave-DMAs)
constexpr int SD_WSB = SD_WPIECES * 16;
constexpr int SD_XROUNDS = 2;                      // activation DMA rounds per stage (64 WAVES slots each)
constexpr int SD_TROUNDS = 4;                      // transposing rounds per stage (4 WAVES units of 8 ch x 16 px each)
constexpr int SD_MAX_LDS8 = 160 * 1024 - 1024;     // 8 waves: one workgroup per CU (the kernel also has 256 bytes of static LDS)
constexpr int SD_MAX_LDS4 = 80 * 1024 - 512;       // 4 waves: two workgroups per CU

struct Sm2dLds {
    int gsz;        // pixels of the halo image (GI x IHT x IWP)
    int xsb;        // bytes of one pixel-major buffer [2 halves][gsz][16] + 1 KB of per-lane trash slots
    int sxp_off;    // per-pixel channel sums
    int xsl;        // 16-byte slots per (image, channel) row of the natural-order buffer
    int xnb;        // bytes of one natural-order buffer (whole DMA rounds)
    int xn_off, ws_off, total;
};
__host__ __device__ inline Sm2dLds sm2d_layout(int GI, int IHT, int IWP, int W, int waves, int wring)
{
    const int SD_THREADS = 64 * waves;
    Sm2dLds L;
    L.gsz = GI * IHT * IWP;
    L.xsb = 2 * L.gsz * 16 + 1024;
    L.sxp_off = 2 * L.xsb;
    L.xsl = (IHT * W + 15) / 16;
    const int slots = GI * 32 * L.xsl;
    L.xnb = (slots + SD_THREADS - 1) / SD_THREADS * SD_THREADS * 16;
    L.xn_off = (L.sxp_off + L.gsz * 4 + 15) & ~15;
    L.ws_off = L.xn_off + 2 * L.xnb;
    L.total = L.ws_off + wring * SD_WSB;
    return L;
}

// 4 transposed reads (8 channels x 16 pixels per 16-lane group each) and their wait in ONE statement: hipcc may copy an
// asm output before a wait that sits in a later statement
__device__ __forceinline__ void sd_tr_read4(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, v2i &r0, v2i &r1, v2i &r2, v2i &r3)
{
    asm volatile("ds_read_b64_tr_b8 %0, %4\n\tds_read_b64_tr_b8 %1, %5\n\tds_read_b64_tr_b8 %2, %6\n\t"
                 "ds_read_b64_tr_b8 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
}
__device__ __forceinline__ void sd_write_b64(uint32_t addr, v2i v)
{
    asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(v) : "memory");
}

// WAVES = 8, WRING = 2: one workgroup per CU, weights double-buffered.  WAVES = 4, WRING = 1: two workgroups per CU (<= 80 KB
// of LDS each), a single weight buffer refilled between two MFMA phases -- its latency is the other workgroup's MFMA time.
template <int NIW, int WAVES, int WRING>
__global__ __launch_bounds__(64 * WAVES, 2) void conv_mfma_sm2d_kernel(const MfmaArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int MT = SD_MT, WN = WAVES / 2, KKT = 9, KW_T = 3;
    constexpr int SD_THREADS = 64 * WAVES, SD_WAVES = WAVES;
    constexpr int WR = (36 + WAVES - 1) / WAVES;        // weight wave-DMAs per wave and stage (the last one partly masked for 8 waves)
    const Sm2dLds L = sm2d_layout(a.GI, a.IHT, a.IWP, a.W, WAVES, WRING);
    const int GSZ = L.gsz;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wms = wave & 1, wn = wave >> 1;
    const int col = lane & 31, h = lane >> 5;

    int pt, ot, th;
    TileGeom g;
    if (!decode_tile(a, pt, ot, g, th)) return;
    const int NT = g.NT;
    const int ih0 = g.oh0 * a.stride - a.pad;
    const int ISZ = a.IHT * a.IWP;
    const int HW = a.H * a.W;
    const int ih_lo = max(ih0, 0), ih_hi = min(ih0 + a.IHT, a.H);
    const int Lb = (ih_hi - ih_lo) * a.W;        // bytes of a channel plane this tile reads (whole rows, contiguous)
    const int nsl = (Lb + 15) >> 4;              // slots that hold them (<= L.xsl)

    // zero both halo images (their borders stay zero = operand 0), the trash slots and the channel sums
    {
        uint4 *z = reinterpret_cast<uint4 *>(smem);
        for (int i = tid; i < (2 * L.xsb) / 16; i += SD_THREADS) z[i] = make_uint4(0, 0, 0, 0);
        int *sx0 = reinterpret_cast<int *>(smem + L.sxp_off);
        for (int i = tid; i < GSZ; i += SD_THREADS) sx0[i] = 0;
    }
    int *sxp = reinterpret_cast<int *>(smem + L.sxp_off);
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;

    // fragment pixel of column slot t (uint4 index inside a pixel-major buffer, half h)
    int pixidx[NIW];
#pragma unroll
    for (int t = 0; t < NIW; ++t) {
        const int q = (wn + t * WN) * 32 + col;
        const int gi = (a.GI > 1) ? q / g.OHWt : 0;
        const int rq = q - gi * g.OHWt;
        const int r = rq / a.OW, c = rq - r * a.OW;
        pixidx[t] = h * GSZ + ((q < NT) ? gi * ISZ + r * a.IWP + c : 0);
    }

    // ---- activation DMA: slot sidx = (round * 8 + wave) * 64 + lane <-> (image gi, channel c of the stage, piece j) ----
    const uint8_t *xi = a.x + (int64_t)g.n0 * a.IC * HW;
    uint32_t xsrc[SD_XROUNDS];
#pragma unroll
    for (int r = 0; r < SD_XROUNDS; ++r) {
        const int sidx = (r * SD_WAVES + wave) * 64 + lane;
        int gi = sidx / (32 * L.xsl);
        const int rem = sidx - gi * (32 * L.xsl);
        int c = rem / L.xsl, j = rem - c * L.xsl;
        if (gi >= a.GI) { gi = 0; c = 0; j = 0; }           // slack of the last round
        if (g.n0 + gi >= a.N) gi = 0;                       // ragged last tile: any valid image (its columns are never stored)
        xsrc[r] = (uint32_t)((gi * a.IC + c) * HW + ih_lo * a.W + min(16 * j, Lb - 16));
    }
    const int x_rounds = L.xnb / (SD_THREADS * 16);        // 1 or 2 (workgroup-uniform)

    // ---- weight DMA: wave-DMA d = round * 8 + wave < 36 <-> segment (tap, half) = d / 2, rows 64 (d & 1) .. + 63 ----
    uint32_t wsrc[WR];
#pragma unroll
    for (int i = 0; i < WR; ++i) {
        const int d = min(i * SD_WAVES + wave, 35);
        const int seg = d >> 1, tap = seg >> 1, hh = seg & 1;
        wsrc[i] = (uint32_t)((((int64_t)tap * a.NG + hh) * a.OCP + ot * MT + (d & 1) * 64 + lane) * 16);
    }
    const uint32_t w_stage = (uint32_t)(2 * a.OCP * 16);    // two 16-channel groups per stage

    // ---- transposing pass: unit u = (round * 8 + wave) * 4 + lane / 16 <-> (image gi, 8-channel group cg, piece j) ----
    int tr_rd[SD_TROUNDS], tr_wr[SD_TROUNDS], tr_sx[SD_TROUNDS];
#pragma unroll
    for (int k = 0; k < SD_TROUNDS; ++k) {
        const int u = (k * SD_WAVES + wave) * 4 + (lane >> 4);
        const int i = lane & 15;
        int gi = u / (4 * L.xsl);
        const int rem = u - gi * (4 * L.xsl);
        int cg = rem / L.xsl, j = rem - cg * L.xsl;
        const bool uok = gi < a.GI && j < nsl;
        if (!uok) { gi = 0; cg = 0; j = 0; }
        tr_rd[k] = ((gi * 32 + cg * 8 + (i >> 1)) * L.xsl + j) * 16 + 8 * (i & 1);
        const int b = (j < nsl - 1) ? 16 * j + i : Lb - 16 + i;       // byte of the row range = pixel
        const bool dup = j == nsl - 1 && b < 16 * (nsl - 1);          // already delivered by the piece before
        const int r = b / a.W, c = b - r * a.W;
        const int idx = (gi * a.IHT + (ih_lo - ih0) + r) * a.IWP + c + a.pad;
        const bool ok = uok && !dup;
        tr_wr[k] = ok ? ((cg >> 1) * GSZ + idx) * 16 + (cg & 1) * 8 : 2 * GSZ * 16 + lane * 16;
        tr_sx[k] = ok ? idx : -1;
    }
    const int t_rounds = (a.GI * 4 * L.xsl + 4 * WAVES - 1) / (4 * WAVES);      // <= SD_TROUNDS (workgroup-uniform)

    int zw_local = 0;
    if (tid < MT) zw_local = (a.ep[a.OCP + ot * MT + tid] != 0.0f) ? 1 : 0;
    const bool need_sx = __syncthreads_or(zw_local) != 0;   // also: zeroing done before any DMA / transposed write

#ifdef QE_STAMP
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = qe_stamp();
    const unsigned long long tstart = tprev;
#endif
    auto issue_x = [&](int s, int buf) __attribute__((always_inline)) {
        const uint8_t *xs = xi + (int64_t)s * 32 * HW;
        uint8_t *dst = smem + L.xn_off + buf * L.xnb + wave * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(xs + xsrc[0]),
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        if (x_rounds > 1)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(xs + xsrc[1]),
                                             (__attribute__((address_space(3))) void *)(dst + SD_WAVES * 1024), 16, 0, 0);
    };
    auto issue_w = [&](int s, int buf) __attribute__((always_inline)) {
        const int8_t *ws = a.wt + (int64_t)s * w_stage;
        uint8_t *dst = smem + L.ws_off + buf * SD_WSB + wave * 1024;
#pragma unroll
        for (int i = 0; i < WR; ++i)
            if (36 % WAVES == 0 || i < WR - 1 || wave < 36 % WAVES)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ws + wsrc[i]),
                                                 (__attribute__((address_space(3))) void *)(dst + i * SD_WAVES * 1024), 16, 0, 0);
    };
    int sxa[SD_TROUNDS] = {0, 0, 0, 0};
    auto transpose_x = [&](int buf) __attribute__((always_inline)) {
        const uint32_t rd0 = lds0 + L.xn_off + buf * L.xnb, wr0 = lds0 + buf * L.xsb;
        v2i v[SD_TROUNDS];
        sd_tr_read4(rd0 + tr_rd[0], rd0 + tr_rd[1], rd0 + tr_rd[2], rd0 + tr_rd[3], v[0], v[1], v[2], v[3]);
#pragma unroll
        for (int k = 0; k < SD_TROUNDS; ++k) {
            if (k < t_rounds) {
                v[k][0] ^= (int)0x80808080u;                 // u - 128 per byte: signed q, or unsigned q - 128
                v[k][1] ^= (int)0x80808080u;
                sd_write_b64(wr0 + tr_wr[k], v[k]);
                if (need_sx && tr_sx[k] >= 0)
                    sxa[k] = __builtin_amdgcn_sdot4(v[k][1], 0x01010101, __builtin_amdgcn_sdot4(v[k][0], 0x01010101, sxa[k], false), false);
            }
        }
    };

    v16i acc0[NIW], acc1[NIW];
#pragma unroll
    for (int t = 0; t < NIW; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[t][r] = 0; acc1[t][r] = 0; }
    }
    auto mma_chunk = [&](int buf, int wbuf) __attribute__((always_inline)) {
        const uint4 *Xs = reinterpret_cast<const uint4 *>(smem + buf * L.xsb);
        const uint4 *Wf = reinterpret_cast<const uint4 *>(smem + L.ws_off + wbuf * SD_WSB) + h * MT + (2 * wms) * 32 + col;
        // fragments of tap k+1 are requested before the MFMAs of tap k (one tap of lookahead)
        v4i af0, af1, b[NIW];
        auto fetch = [&](int tap, v4i &f0, v4i &f1, v4i (&bb)[NIW]) __attribute__((always_inline)) {
            const int off = (tap / KW_T) * a.IWP + (tap % KW_T);
            const uint4 w0 = Wf[tap * 2 * MT], w1 = Wf[tap * 2 * MT + 32];
            f0 = v4i{(int)w0.x, (int)w0.y, (int)w0.z, (int)w0.w};
            f1 = v4i{(int)w1.x, (int)w1.y, (int)w1.z, (int)w1.w};
#pragma unroll
            for (int t = 0; t < NIW; ++t) bb[t] = *reinterpret_cast<const v4i *>(&Xs[pixidx[t] + off]);
        };
        fetch(0, af0, af1, b);
#pragma unroll
        for (int tap = 0; tap < KKT; ++tap) {
            v4i nf0 = af0, nf1 = af1, nb[NIW];
#pragma unroll
            for (int t = 0; t < NIW; ++t) nb[t] = b[t];
            if (tap + 1 < KKT) fetch(tap + 1, nf0, nf1, nb);
#pragma unroll
            for (int t = 0; t < NIW; ++t) {
                acc0[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af0, b[t], acc0[t], 0, 0, 0);
                acc1[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af1, b[t], acc1[t], 0, 0, 0);
            }
            af0 = nf0; af1 = nf1;
#pragma unroll
            for (int t = 0; t < NIW; ++t) b[t] = nb[t];
            // one fragment read of tap k+1 ahead of each of the first MFMAs of tap k
#pragma unroll
            for (int j = 0; j < NIW + 2; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
            }
            if constexpr (2 * NIW > NIW + 2) __builtin_amdgcn_sched_group_barrier(0x008, 2 * NIW - (NIW + 2), 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if constexpr (WRING == 2) {
        auto stage = [&](int s, auto pf_w, auto pf_x, auto tr) __attribute__((always_inline)) {
            if constexpr (decltype(pf_w)::value) issue_w(s + 1, (s + 1) & 1);
            if constexpr (decltype(pf_x)::value) issue_x(s + 2, s & 1);
            QE_ST(3);   // DMA issue
            if constexpr (decltype(tr)::value) transpose_x((s + 1) & 1);
            QE_ST(1);   // transposing pass
            mma_chunk(s & 1, s & 1);
            QE_ST(4);   // MFMA phase
            __syncthreads();   // vmcnt(0): this wave's DMAs of the stage have landed; lgkmcnt(0): its transposed writes too
            QE_ST(2);
        };
        issue_w(0, 0);
        issue_x(0, 0);
        issue_x(1, 1);
        __syncthreads();
        transpose_x(0);
        __syncthreads();
        QE_ST(0);
        for (int s = 0; s < a.NCH - 2; ++s) stage(s, std::true_type{}, std::true_type{}, std::true_type{});
        stage(a.NCH - 2, std::true_type{}, std::false_type{}, std::true_type{});
        stage(a.NCH - 1, std::false_type{}, std::false_type{}, std::false_type{});
    } else {
        // single weight buffer: W(s) is requested when the last MFMA of stage s-1 is done (barrier B) and waited for with a
        // counted vmcnt that leaves the younger X(s+2) requests in flight; the other workgroup of the CU computes meanwhile
        auto stage = [&](int s, auto pf_x, auto tr) __attribute__((always_inline)) {
            issue_w(s, 0);
            if constexpr (decltype(pf_x)::value) issue_x(s + 2, s & 1);
            QE_ST(3);   // DMA issue
            if constexpr (decltype(tr)::value) transpose_x((s + 1) & 1);
            QE_ST(1);   // transposing pass
            if constexpr (decltype(pf_x)::value) {
                if (x_rounds > 1) __builtin_amdgcn_s_waitcnt(0x0f72);   // vmcnt(2)
                else __builtin_amdgcn_s_waitcnt(0x0f71);                // vmcnt(1)
            } else
                __builtin_amdgcn_s_waitcnt(0x0f70);                     // vmcnt(0)
            __builtin_amdgcn_s_barrier();   // barrier A: W(s) complete
            QE_ST(2);
            mma_chunk(s & 1, 0);
            QE_ST(4);   // MFMA phase
            __syncthreads();   // barrier B: X(s+2) landed, transposed X(s+1) visible, Ws free
            QE_ST(5);
        };
        issue_x(0, 0);
        issue_x(1, 1);
        __syncthreads();
        transpose_x(0);
        __syncthreads();
        QE_ST(0);
        for (int s = 0; s < a.NCH - 2; ++s) stage(s, std::true_type{}, std::true_type{});
        stage(a.NCH - 2, std::false_type{}, std::true_type{});
        stage(a.NCH - 1, std::false_type{}, std::false_type{});
    }

    if (need_sx) {
#pragma unroll
        for (int k = 0; k < SD_TROUNDS; ++k)
            if (k < t_rounds && tr_sx[k] >= 0) atomicAdd(&sxp[tr_sx[k]], sxa[k]);
        __syncthreads();
    }
    int sxs[NIW];
#pragma unroll
    for (int t = 0; t < NIW; ++t) {
        sxs[t] = 0;
        if (need_sx) {
            const int pbase = pixidx[t] - h * GSZ;
            for (int tap = 0; tap < KKT; ++tap) sxs[t] += sxp[pbase + (tap / KW_T) * a.IWP + (tap % KW_T)];
        }
    }
    QE_ST(6);
    const float *ctab = stage_ctab<MT>(a, smem, ot, tid, SD_THREADS);
    const int *ptab = ctab ? nullptr : stage_ptab<MT>(a, smem, ot, tid, SD_THREADS);
    mfma_epilogue<4, WN, NIW>(a, acc0, sxs, need_sx, g, ot, 2 * wms, wn, col, h, KKT, ptab, ctab);
    mfma_epilogue<4, WN, NIW>(a, acc1, sxs, need_sx, g, ot, 2 * wms + 1, wn, col, h, KKT, ptab, ctab);
#ifdef QE_STAMP
    QE_ST(7);
    if (a.dbg != nullptr && lane == 0) {
        unsigned long long *o = a.dbg + ((size_t)blockIdx.x * 4 + (wave & 3)) * 10;   // (the tool assumes 4 waves per block)
        if (wave < 4) {
            for (int i = 0; i < 8; ++i) o[i] = st[i];
            o[8] = tprev - tstart;
            o[9] = tstart;
        }
    }
#endif
}

// Plan of the sm2d tiling for a 3x3 / stride 1 / pad 1 layer: GI whole images (small planes) or a band of TH output rows
// per workgroup, <= 32 * (waves / 2) * 4 pixels.  Returns false when the layer does not fit this kernel.
bool sm2d_plan(const qe_conv_shape *sh, int waves, int *GI, int *TH, int *niw, size_t *lds)
{
    if (sh->KH != 3 || sh->KW != 3 || sh->stride != 1 || sh->padding != 1) return false;
    if (sh->OC <= 64 || (sh->IC % 32) != 0 || sh->IC < 64) return false;
    const int OH = sh->H, OW = sh->W, P = OH * OW;
    const int n_oc = (sh->OC + SD_MT - 1) / SD_MT;
    const int max_px = 32 * (waves / 2) * 4, threads = 64 * waves, wring = waves == 8 ? 2 : 1;
    const int max_lds = waves == 8 ? SD_MAX_LDS8 : SD_MAX_LDS4;
    int gi = 1, th = OH;
    if (P <= max_px / 2) {
        // whole images: as many as the pixel slots hold, but keep >= 256 workgroups where the batch allows
        gi = std::max(1, std::min((int)sh->N, max_px / P));
        while (gi > 1 && (int64_t)((sh->N + gi - 1) / gi) * n_oc < 256) --gi;
    } else if (P > max_px) {
        th = std::min(OH, max_px / OW);
        if (th < 1) return false;
        const int nt = (OH + th - 1) / th;
        th = (OH + nt - 1) / nt;            // balanced row bands
    }
    for (;;) {
        const int IHT = th + 2, IWP = OW + 2;
        const Sm2dLds L = sm2d_layout(gi, IHT, IWP, sh->W, waves, wring);
        const bool rows_ok = sh->W >= 8 || (th == OH && P >= 16);       // every tile reads >= 16 contiguous bytes per channel
        if (rows_ok && gi * 32 * L.xsl <= SD_XROUNDS * threads && gi * 4 * L.xsl <= SD_TROUNDS * 4 * waves && L.total <= max_lds &&
            (int64_t)gi * sh->IC * sh->H * sh->W < (1ll << 31)) {
            *GI = gi; *TH = th; *niw = (gi * th * OW <= max_px / 2) ? 2 : 4; *lds = (size_t)L.total;
            return true;
        }
        if (gi > 1) { --gi; continue; }
        if (--th < 1) return false;
    }
}

int sm2d_ptab_off(int GI, int IHT, int IWP, int W, int waves) { return sm2d_layout(GI, IHT, IWP, W, waves, waves == 8 ? 2 : 1).xn_off; }

template <int NIW, int WAVES>
static void launch_one(const MfmaArgs &a, unsigned blocks, size_t lds, hipStream_t s)
{
    constexpr int WRING = WAVES == 8 ? 2 : 1;
    static const bool raised = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma_sm2d_kernel<NIW, WAVES, WRING>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   WAVES == 8 ? SD_MAX_LDS8 : SD_MAX_LDS4) == hipSuccess;
    (void)raised;
    hipLaunchKernelGGL((conv_mfma_sm2d_kernel<NIW, WAVES, WRING>), dim3(blocks), dim3(64 * WAVES), lds, s, a);
}

void launch_mfma_sm2d(const MfmaArgs &a, int niw, int waves, unsigned blocks, size_t lds, hipStream_t s)
{
    if (waves == 8) {
        if (niw == 2) launch_one<2, 8>(a, blocks, lds, s);
        else launch_one<4, 8>(a, blocks, lds, s);
    } else {
        if (niw == 2) launch_one<2, 4>(a, blocks, lds, s);
        else launch_one<4, 4>(a, blocks, lds, s);
    }
}

}  // namespace qe
