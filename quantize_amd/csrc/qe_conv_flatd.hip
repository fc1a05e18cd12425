// qe_conv_flatd.hip -- 1x1 / stride 1 / pad 0 quantconv2d as a GEMM whose operand stages arrive by LDS-DMA ("flatd").
//
// Replaces the per-element loop of quantconv2d_cuda_kernel (engine/kernels/functions/quantconv2d.cu:78-141) for
// 8-bit x 8-bit 1x1 layers with IC % 64 == 0.  Same arithmetic as conv_mfma_flat_kernel (qe_conv_mfma_kernel.hpp):
//   out[n,oc,p] = bias[oc] + sx sw[oc] ( S_aw - zw' S_x - zx' S_w + IC zx' zw' ),  a = u ^ 0x80, S_aw on
//   v_mfma_i32_32x32x32_i8 with A = activations (rows = pixels, fragments by ds_read_b64_tr_b8 from the native
//   [channel][pixel] order), B = weights (cols = output channel, 16 contiguous bytes of the packed OIHW row).
// What differs is how the operands reach LDS.  The register-staged flat kernel keeps ONE stage of loads in flight per
// workgroup (its staging registers), and a layer with a long K loop and small planes (1024->256 @14x14, 2048->512 @7x7)
// pays one memory round trip per stage: 2.8 TB/s and 1.0 TB/s in the ResNet-50 stack, bound by neither HBM nor MFMA.
// Here both operands of a 64-channel stage are written into a ring of 3 LDS buffers by global_load_lds_dwordx4 (no
// staging registers, nothing to transpose or convert on the way), two stages ahead of the MFMAs, with one raw
// s_barrier and a counted vmcnt per stage (a __syncthreads() would drain the DMA queue).  The u ^ 0x80 recode moves to
// the fragments.
//
// LDS image of the activations = what ds_read_b64_tr_b8 wants: rows = channels, 16-pixel groups at 8-byte aligned
// addresses, row stride an odd multiple of 32 B (conflict-free).  LDS-DMA writes lane-linear 16-byte slots, but the
// GLOBAL address is per lane, so slot (channel c, j) simply fetches plane bytes [16 j, 16 j + 16) of channel c:
//   WIDE  (planes >= 160 pixels; tile = 32 NT pixels of one plane, NT = 5 | 7): row stride 32 NT bytes, 2 NT slots per row;
//         a 14x14 plane (196 B) is one 224-pixel tile whose rows start 4-byte aligned in global memory -- LDS-DMA takes
//         that (tools/probe_dma_align.hip) -- and whose last 28 bytes per row are the next channel's (never stored);
//   SMALL (planes <= 64 pixels: 7x7; tile = 4 whole images x 2 column tiles): 64-byte rows [image][channel][64], byte-aligned
//         sources, slot position XOR 2 for channels with bit 2 set so that the 8 rows x 2 pixel groups of a transposed read
//         cover all 64 banks once.
// The only bytes such slots could read past the tensor are those behind the LAST plane of the LAST image when the plane
// size is not a multiple of 16: pure-garbage slots fetch the tensor's last 16 bytes instead, and the one slot that is
// partly valid is left out of the DMA (lane masked off) and written by its thread from a register.
#include "qe_conv_mfma_kernel.hpp"

#include <cstdlib>
#include <utility>

namespace qe {

struct FlatdArgs {
    const uint8_t *x;          // [N][IC][P] stored codes, 8-bit
    const uint8_t *w;          // [OC][IC] stored codes, 8-bit
    const float *x_scale, *x_zero, *w_scale, *w_zero, *bias;
    int x_sign, w_sign, w_per_tensor;
    float *out;                // [N][OC][P] fp32
    int N, IC, OC, P;
    int tiles_per_image;       // WIDE: pixel tiles per plane; SMALL: unused
    int n_pix_tiles, n_oc_tiles, chunk;
    // fused re-quantisation (7x7 planes, RQ instances): the 8-bit code of the consumer's quantiser instead of fp32, fields as in MfmaArgs
    uint8_t *rq_out;
    const float *rq_scale, *rq_zero;
    float rq_qmin, rq_qmax, rq_lo, rq_hi;
    unsigned rq_offset;
    int32_t *rq_status;
};

constexpr int FD_CK = 64;              // channels per stage
constexpr int FD_RING_DEFAULT = 3;      // slots of the ring (RING - 1 stages in flight); 6 = one workgroup per CU with 5 in flight
// output channels per workgroup = 32 per wave: 4 waves (128 channels, two workgroups per CU) or 8 waves (256 channels, one
// workgroup per CU: the activations of a pixel tile cross the CU's memory path once per 256 output channels instead of
// once per 128 -- the bytes that bound these layers)

template <int NT, bool SMALL, int RING = FD_RING_DEFAULT, int WAVES = 4> struct FdGeom {
    static constexpr int MT = 32 * WAVES;
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int FD_WBYTES = MT * FD_CK;
    static constexpr int RS = SMALL ? 64 : 32 * NT;                  // LDS bytes per channel row (WIDE: NT odd)
    static constexpr int XBYTES = SMALL ? 4 * FD_CK * 64 : FD_CK * RS;
    static constexpr int STAGE = XBYTES + FD_WBYTES;
    static constexpr int XINSTR = XBYTES / 1024;                     // wave-level DMA instructions per stage
    static constexpr int NTP = 32 * NT;
    static constexpr int PATCH = SMALL ? WAVES * 32 * 49 * 4 : WAVES * 32 * 36 * 4;
    static constexpr int RING_BYTES = RING * STAGE > PATCH ? RING * STAGE : PATCH;
    static constexpr int LDS = RING_BYTES + WAVES * NTP * 4 /* S_x, one copy per wave */;
};

// ---------------------------------------------------------------------------------------------
// LDS reads as inline asm.  hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of every transposed-read builtin (and of
// any plain LDS store) that follows an LDS-DMA: their memory operands carry no alias information, so the wait-count
// pass assumes they touch what the DMA writes -- which drained the stages in flight once per stage.  The ring protocol
// (counted vmcnt + barrier) already orders these accesses behind the DMA that fills their slot, so they are written by
// hand.  Rule kept throughout: an asm statement's outputs are VALID when the statement ends (the reads and their
// `s_waitcnt lgkmcnt(0)` live in ONE statement).  Leaving the wait to a later statement is not safe: hipcc may copy an
// output register right behind the statement that declares it, i.e. before the data has arrived (seen: v_mov of
// not-yet-loaded patch rows -> garbage in rows 16-31 of every tile).
// ---------------------------------------------------------------------------------------------
// all fragment reads of one 32-channel chunk of a WIDE stage: weight fragment + NT x (8 + 8 channels) transposed reads
template <int NT> __device__ __forceinline__ void fd_reads_wide(uint32_t tr_addr, uint32_t wf_addr, v4i &wf, v2i (&lo)[NT], v2i (&hi)[NT]);
template <> __device__ __forceinline__ void fd_reads_wide<5>(uint32_t tr_addr, uint32_t wf_addr, v4i &wf, v2i (&lo)[5], v2i (&hi)[5])
{
    asm volatile(
        "ds_read_b128 %0, %12\n\t"
        "ds_read_b64_tr_b8 %1, %11 offset:0\n\t"
        "ds_read_b64_tr_b8 %6, %11 offset:1280\n\t"
        "ds_read_b64_tr_b8 %2, %11 offset:32\n\t"
        "ds_read_b64_tr_b8 %7, %11 offset:1312\n\t"
        "ds_read_b64_tr_b8 %3, %11 offset:64\n\t"
        "ds_read_b64_tr_b8 %8, %11 offset:1344\n\t"
        "ds_read_b64_tr_b8 %4, %11 offset:96\n\t"
        "ds_read_b64_tr_b8 %9, %11 offset:1376\n\t"
        "ds_read_b64_tr_b8 %5, %11 offset:128\n\t"
        "ds_read_b64_tr_b8 %10, %11 offset:1408\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(wf), "=&v"(lo[0]), "=&v"(lo[1]), "=&v"(lo[2]), "=&v"(lo[3]), "=&v"(lo[4]), "=&v"(hi[0]), "=&v"(hi[1]), "=&v"(hi[2]), "=&v"(hi[3]), "=&v"(hi[4])
        : "v"(tr_addr), "v"(wf_addr));
}
template <> __device__ __forceinline__ void fd_reads_wide<7>(uint32_t tr_addr, uint32_t wf_addr, v4i &wf, v2i (&lo)[7], v2i (&hi)[7])
{
    asm volatile(
        "ds_read_b128 %0, %16\n\t"
        "ds_read_b64_tr_b8 %1, %15 offset:0\n\t"
        "ds_read_b64_tr_b8 %8, %15 offset:1792\n\t"
        "ds_read_b64_tr_b8 %2, %15 offset:32\n\t"
        "ds_read_b64_tr_b8 %9, %15 offset:1824\n\t"
        "ds_read_b64_tr_b8 %3, %15 offset:64\n\t"
        "ds_read_b64_tr_b8 %10, %15 offset:1856\n\t"
        "ds_read_b64_tr_b8 %4, %15 offset:96\n\t"
        "ds_read_b64_tr_b8 %11, %15 offset:1888\n\t"
        "ds_read_b64_tr_b8 %5, %15 offset:128\n\t"
        "ds_read_b64_tr_b8 %12, %15 offset:1920\n\t"
        "ds_read_b64_tr_b8 %6, %15 offset:160\n\t"
        "ds_read_b64_tr_b8 %13, %15 offset:1952\n\t"
        "ds_read_b64_tr_b8 %7, %15 offset:192\n\t"
        "ds_read_b64_tr_b8 %14, %15 offset:1984\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(wf), "=&v"(lo[0]), "=&v"(lo[1]), "=&v"(lo[2]), "=&v"(lo[3]), "=&v"(lo[4]), "=&v"(lo[5]), "=&v"(lo[6]), "=&v"(hi[0]), "=&v"(hi[1]), "=&v"(hi[2]), "=&v"(hi[3]), "=&v"(hi[4]), "=&v"(hi[5]), "=&v"(hi[6])
        : "v"(tr_addr), "v"(wf_addr));
}
// SMALL: tile T = image T / 2 (4096 bytes apart), column tile T % 2 (slot position differs by XOR 2: second base)
__device__ __forceinline__ void fd_reads_small(uint32_t tr_addr0, uint32_t tr_addr1, uint32_t wf_addr, v4i &wf, v2i (&lo)[8], v2i (&hi)[8])
{
    asm volatile(
        "ds_read_b128 %0, %19\n\t"
        "ds_read_b64_tr_b8 %1, %17 offset:0\n\t"
        "ds_read_b64_tr_b8 %9, %17 offset:512\n\t"
        "ds_read_b64_tr_b8 %2, %18 offset:0\n\t"
        "ds_read_b64_tr_b8 %10, %18 offset:512\n\t"
        "ds_read_b64_tr_b8 %3, %17 offset:4096\n\t"
        "ds_read_b64_tr_b8 %11, %17 offset:4608\n\t"
        "ds_read_b64_tr_b8 %4, %18 offset:4096\n\t"
        "ds_read_b64_tr_b8 %12, %18 offset:4608\n\t"
        "ds_read_b64_tr_b8 %5, %17 offset:8192\n\t"
        "ds_read_b64_tr_b8 %13, %17 offset:8704\n\t"
        "ds_read_b64_tr_b8 %6, %18 offset:8192\n\t"
        "ds_read_b64_tr_b8 %14, %18 offset:8704\n\t"
        "ds_read_b64_tr_b8 %7, %17 offset:12288\n\t"
        "ds_read_b64_tr_b8 %15, %17 offset:12800\n\t"
        "ds_read_b64_tr_b8 %8, %18 offset:12288\n\t"
        "ds_read_b64_tr_b8 %16, %18 offset:12800\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(wf), "=&v"(lo[0]), "=&v"(lo[1]), "=&v"(lo[2]), "=&v"(lo[3]), "=&v"(lo[4]), "=&v"(lo[5]), "=&v"(lo[6]), "=&v"(lo[7]), "=&v"(hi[0]), "=&v"(hi[1]), "=&v"(hi[2]), "=&v"(hi[3]), "=&v"(hi[4]), "=&v"(hi[5]), "=&v"(hi[6]), "=&v"(hi[7])
        : "v"(tr_addr0), "v"(tr_addr1), "v"(wf_addr));
}
// the 4 row groups of a wave's 32 x 36 float store patch (8 rows = 1152 bytes apart)
__device__ __forceinline__ void fd_patch_read4(uint32_t addr, v4i &o0, v4i &o1, v4i &o2, v4i &o3)
{
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1152\n\tds_read_b128 %2, %4 offset:2304\n\t"
                 "ds_read_b128 %3, %4 offset:3456\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3) : "v"(addr) : "memory");
}
__device__ __forceinline__ void fd_patch_write(uint32_t lds_addr, float a, float b, float c, float d)
{
    const v4i v = {__float_as_int(a), __float_as_int(b), __float_as_int(c), __float_as_int(d)};
    asm volatile("ds_write_b128 %0, %1" : : "v"(lds_addr), "v"(v) : "memory");
}
__device__ __forceinline__ void fd_lds_drain() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ void fd_wait_vmcnt(int n)   // n wave-uniform, 0..31
{
#define QE_VMW(k) case k: __builtin_amdgcn_s_waitcnt(0x0f70 | ((k) & 15) | (((k) >> 4) << 14)); break;
    switch (n) {
        QE_VMW(0) QE_VMW(1) QE_VMW(2) QE_VMW(3) QE_VMW(4) QE_VMW(5) QE_VMW(6) QE_VMW(7) QE_VMW(8) QE_VMW(9) QE_VMW(10)
        QE_VMW(11) QE_VMW(12) QE_VMW(13) QE_VMW(14) QE_VMW(15) QE_VMW(16) QE_VMW(17) QE_VMW(18) QE_VMW(19) QE_VMW(20)
        QE_VMW(21) QE_VMW(22) QE_VMW(23) QE_VMW(24) QE_VMW(25) QE_VMW(26) QE_VMW(27) QE_VMW(28) QE_VMW(29) QE_VMW(30)
        default: __builtin_amdgcn_s_waitcnt(0x0f70); break;
    }
#undef QE_VMW
}

template <int NT, bool SMALL, int FD_RING, int WAVES, bool RQ = false>
__global__ __launch_bounds__(64 * WAVES, 2) void conv_flatd_kernel(const FlatdArgs a)
{
    using G = FdGeom<NT, SMALL, FD_RING, WAVES>;
    constexpr int AHEAD = FD_RING - 1;
    constexpr int FD_MT = G::MT, THREADS = G::THREADS;
    constexpr int RS = G::RS;
    static_assert(SMALL || (NT & 1) == 1, "WIDE tiles need an odd tile count (row stride = odd multiple of 32 B)");
    static_assert(!SMALL || NT == 8, "SMALL tiles are 4 images x 2 column tiles");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 31, h = lane >> 5;
    const int P = a.P;

    // ---- tile decode: XCD-aware block map (same scheme as block_to_tile of the register-staged kernels) -------
    int pt, ot;
    {
        const int bid = blockIdx.x;
        const int idx = bid >> 3;
        const int j = idx / a.n_oc_tiles;
        ot = idx - j * a.n_oc_tiles;
        const int c = j / a.chunk;
        pt = (c * 8 + (bid & 7)) * a.chunk + (j - c * a.chunk);
    }
    if (pt >= a.n_pix_tiles) return;
    int n0, p0;
    if constexpr (SMALL) { n0 = pt * 4; p0 = 0; }
    else { n0 = pt / a.tiles_per_image; p0 = (pt - n0 * a.tiles_per_image) * G::NTP; }

    // ---- epilogue constants of this lane's output channel ------------------------------------------------------
    const int oc = ot * FD_MT + wave * 32 + col;
    const int occ = oc < a.OC ? oc : a.OC - 1;
    const float sw = a.w_per_tensor ? a.w_scale[0] : a.w_scale[occ];
    const float zwp = (a.w_per_tensor ? a.w_zero[0] : a.w_zero[occ]) - (a.w_sign ? 0.0f : 128.0f);
    const float alpha = a.x_scale[0] * sw;
    const float bia = a.bias ? a.bias[occ] : 0.0f;
    const float zxp = a.x_zero[0] - (a.x_sign ? 0.0f : 128.0f);

    // ---- DMA sources (stage 0; a stage advances X by 64 planes and W by 64 bytes) ------------------------------
    // X: LDS slot e = 64 * (wave-instruction q) + lane, q = wave, wave + 4, ...
    constexpr int PXW = (G::XINSTR + WAVES - 1) / WAVES;   // X instructions of waves 0 .. (XINSTR % WAVES) - 1 (the others issue one less)
    const int n_xi = (G::XINSTR % WAVES == 0 || wave < G::XINSTR % WAVES) ? PXW : PXW - 1;
    const int64_t x_last16 = (int64_t)a.N * a.IC * P - 16;      // last address a 16-byte read may start at
    const uint8_t *px[PXW];
    int fix_lds = -1, fix_i = -1;                      // this thread writes the tensor's last bytes itself (last stage only)
    uint32_t fix_val = 0;
#pragma unroll
    for (int i = 0; i < PXW; ++i) {
        const int e = 64 * (wave + WAVES * i) + lane;
        int64_t src;
        bool last_row;
        int j;
        if constexpr (SMALL) {
            const int img = e >> 8, c = (e >> 2) & 63, jj = e & 3;
            j = jj ^ (2 * ((c >> 2) & 1));
            const int n = n0 + img < a.N ? n0 + img : a.N - 1;
            src = ((int64_t)n * a.IC + c) * P + 16 * j;
            last_row = (n0 + img == a.N - 1) && c == FD_CK - 1;
        } else {
            const int c = e / (RS / 16);
            j = e - c * (RS / 16);
            src = ((int64_t)n0 * a.IC + c) * P + p0 + 16 * j;
            last_row = (n0 == a.N - 1) && c == FD_CK - 1;
        }
        // stage s adds s * 64 planes: only the last stage of the last image can run past the tensor
        const int64_t lim = x_last16 - (int64_t)(a.IC - FD_CK) * P;
        if (last_row && src > lim) {
            const int pos = (SMALL ? 0 : p0) + 16 * j;            // first plane byte this slot should hold
            if (pos < P && e < G::XBYTES / 16) {
                // partly valid (P - pos = 4 or 1 bytes): in the last stage this lane is masked out of the DMA (see
                // issue()) and stores the valid bytes itself; in every earlier stage the slot is an ordinary in-bounds
                // read, so its address stays as it is
                fix_lds = e * 16;
                fix_i = i;
                uint32_t v = 0;
                const uint8_t *tail = a.x + (int64_t)a.N * a.IC * P - (P - pos);
                for (int b = 0; b < 4 && b < P - pos; ++b) v |= (uint32_t)tail[b] << (8 * b);
                fix_val = v;
            } else {
                src = lim;                                         // pure garbage slot: any valid 16 bytes will do, in every stage
            }
        }
        px[i] = a.x + src;
    }
    // W: slot e = tid + 256 i <-> row e >> 2 of the tile, position e & 3 holds k-piece (e & 3) ^ ((row >> 2) & 3)
    const uint8_t *pw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = tid + THREADS * i;
        const int r = e >> 2, sl = e & 3;
        const int orow = ot * FD_MT + r < a.OC ? ot * FD_MT + r : a.OC - 1;
        pw[i] = a.w + (int64_t)orow * a.IC + 16 * (sl ^ ((r >> 2) & 3));
    }
    const int64_t x_step = (int64_t)FD_CK * P;
    const int n_stages = a.IC / FD_CK;
    auto issue = [&](int s) __attribute__((always_inline)) {
        uint8_t *buf = smem + (s % FD_RING) * G::STAGE;
        // the one slot that would read past the tensor (last stage of the last plane) is left out of the DMA: its lane is
        // masked off and writes the valid bytes with a plain LDS store before the stage's barrier (a store issued after
        // `s_waitcnt vmcnt(0)` on top of a DMA'd slot was observed to lose against the DMA's own LDS write)
        const int skip = (s == n_stages - 1) ? fix_i : -1;
#pragma unroll
        for (int i = 0; i < PXW; ++i) {
            if (i < n_xi && i != skip)    // first term wave-uniform, second per lane (EXEC mask)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(px[i] + s * x_step),
                                                 (__attribute__((address_space(3))) void *)(buf + 1024 * (wave + WAVES * i)), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(pw[i] + s * FD_CK),
                                             (__attribute__((address_space(3))) void *)(buf + G::XBYTES + (THREADS * i + 64 * wave) * 16), 16, 0, 0);
    };

    for (int s0 = 0; s0 < AHEAD && s0 < n_stages; ++s0) issue(s0);

    // S_x (per-pixel channel sums) is only needed by output channels with zw' != 0: decided per WAVE (its 32 channels),
    // so the prologue needs no workgroup barrier (a __syncthreads() here would drain the two stages just requested)
    const bool need_sx = __builtin_amdgcn_ballot_w64(oc < a.OC && zwp != 0.0f) != 0ull;
    int *sxp = reinterpret_cast<int *>(smem + G::RING_BYTES) + wave * G::NTP;

    v16i acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0;
    int swacc = 0;
    int sxacc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) sxacc[t] = 0;

    // fragment addressing (32-bit LDS byte addresses for the hand-written transposed reads)
    const int i16 = lane & 15;
    const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)smem;
    uint32_t tr_b0, tr_b1;
    if constexpr (SMALL) {
        // row (16 h + i16/2) of [image][channel][64]; logical slot 2 tt + (lane>>4)&1 sits at position slot ^ 2 for rows 4-7
        const int sm_sl = ((lane >> 4) & 1) ^ (2 * (i16 >> 3));
        const uint32_t rowb = (uint32_t)((16 * h + (i16 >> 1)) * 64 + 8 * (i16 & 1));
        tr_b0 = rowb + 16u * (uint32_t)sm_sl;
        tr_b1 = rowb + 16u * (uint32_t)(sm_sl ^ 2);
    } else {
        tr_b0 = (uint32_t)((16 * h + (i16 >> 1)) * RS + 16 * ((lane >> 4) & 1) + 8 * (i16 & 1));
        tr_b1 = tr_b0;
    }
    const int wf_base = G::XBYTES + (wave * 32 + col) * FD_CK;
    const int wswz = (col >> 2) & 3;

    auto main_loop = [&](auto sx_tag) __attribute__((always_inline)) {
        constexpr bool SX = decltype(sx_tag)::value;
        for (int s = 0; s < n_stages; ++s) {
            // stage s has landed for this wave once at most the pieces of stage s + 1 are still outstanding
            if (s + 1 < n_stages) {
                const int rem = (AHEAD - 1) < (n_stages - 1 - s) ? (AHEAD - 1) : (n_stages - 1 - s);   // younger stages in flight
                fd_wait_vmcnt(rem * (n_xi + 2));
            } else {
                __builtin_amdgcn_s_waitcnt(0x0f70);
                if (fix_lds >= 0) {
                    uint8_t *dst = smem + (s % FD_RING) * G::STAGE + fix_lds;
                    if constexpr (SMALL) *dst = (uint8_t)fix_val; else *reinterpret_cast<uint32_t *>(dst) = fix_val;
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                }
            }
            __builtin_amdgcn_s_barrier();             // ... for every wave; and ring slot (s + 2) % 3 has no reader left
            if (s + AHEAD < n_stages) issue(s + AHEAD);
            // per 32-channel chunk: one asm statement = weight fragment + every activation fragment + their wait
            const uint32_t xb = smem_lds + (uint32_t)((s % FD_RING) * G::STAGE);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                v2i lo[NT], hi[NT];
                v4i wf;
                const uint32_t wfa = xb + (uint32_t)(wf_base + 16 * ((2 * k + h) ^ wswz));
                if constexpr (SMALL) fd_reads_small(xb + tr_b0 + k * 32 * 64, xb + tr_b1 + k * 32 * 64, wfa, wf, lo, hi);
                else fd_reads_wide<NT>(xb + tr_b0 + k * 32 * RS, wfa, wf, lo, hi);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    wf[j] ^= (int)0x80808080;         // u - 128: signed q, or unsigned q - 128
                    swacc = __builtin_amdgcn_sdot4(wf[j], 0x01010101, swacc, false);
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    v4i xf = {lo[t][0], lo[t][1], hi[t][0], hi[t][1]};
#pragma unroll
                    for (int j = 0; j < 4; ++j) xf[j] ^= (int)0x80808080;
                    if constexpr (SX) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) sxacc[t] = __builtin_amdgcn_sdot4(xf[j], 0x01010101, sxacc[t], false);
                    }
                    acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xf, wf, acc[t], 0, 0, 0);
                }
            }
        }
    };
    if (need_sx) main_loop(std::true_type{}); else main_loop(std::false_type{});

    // ---- epilogue ------------------------------------------------------------------------------------------------
    const int sw_sum = swacc + __shfl_xor(swacc, 32);
    const float cst = fmaf((float)a.IC * zxp, zwp, -zxp * (float)sw_sum);
    if (need_sx) {
        // S_x lives on the lane that owns the pixel as an A row; the accumulators have the pixel on the register:
        // through this wave's own LDS copy (written and read by the same wave)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int v = sxacc[t] + __shfl_xor(sxacc[t], 32);
            if (h == 0) sxp[32 * t + col] = v;
        }
    }
    __syncthreads();          // every wave is done with the ring (it becomes the store patches)
    if constexpr (!SMALL) {
        // each wave turns its 32 oc x 32 px tiles through a private patch: one global_store_dwordx4 = 8 rows x 128 B
        const int NTv = min(G::NTP, P - p0);
        float *out_w = a.out + ((int64_t)n0 * a.OC + ot * FD_MT + wave * 32) * P + p0;
        float *patch = reinterpret_cast<float *>(smem) + wave * (32 * 36);
        const int rrow = lane >> 3, rq = lane & 7;
        const uint32_t voff = (uint32_t)rrow * (uint32_t)P + 4u * (uint32_t)rq;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
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
            __builtin_amdgcn_s_waitcnt(0xc07f);
            const bool px_ok = q0 + 4 * rq < NTv;
            if (q0 + 32 <= NTv && ot * FD_MT + wave * 32 + 32 <= a.OC) {
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
                    if (px_ok && ot * FD_MT + wave * 32 + row < a.OC)
                        *reinterpret_cast<float4 *>(out_w + (int64_t)(8 * i) * P + q0 + voff) = o4;
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
        }
    } else if constexpr (RQ) {
        // fused re-quantisation: a wave's 32 channels x 49 codes of one image are ONE contiguous 1,568-byte run of the output
        // (16-byte aligned: oc0 % 32 == 0 and OC % 32 == 0, host) -- laid out in the patch as in memory, copied flat.  The
        // fp32 value is computed exactly as below, then quantised as quantize_pack would (rq_value / rq_fast2).
        static_assert(SMALL, "re-quantising instances exist for the 7x7 form only");
        uint8_t *bp = smem + wave * (32 * 49 * 4);
        const int oc0 = ot * FD_MT + wave * 32;
        RqConst rqc = rq_setup(a);
        rqc.slow = __builtin_amdgcn_readfirstlane(rqc.slow);
        rqc.chk = __builtin_amdgcn_readfirstlane(rqc.chk);
        bool bad = false;
        const bool fast = rq_fast_ok(rqc) && __builtin_amdgcn_ballot_w64(!rq_bounded(alpha, cst, bia, zwp)) == 0ull;
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
            const bool img = n0 + gi < a.N;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int pxl = 32 * tt + (r & 3) + 8 * (r >> 2) + 4 * h;      // r even: pxl, pxl + 1 are registers r, r + 1
                    float f0 = (float)acc[2 * gi + tt][r] + cst, f1 = (float)acc[2 * gi + tt][r + 1] + cst;
                    if (need_sx) {
                        f0 = fmaf(-zwp, (float)sxp[64 * gi + pxl], f0);
                        f1 = fmaf(-zwp, (float)sxp[64 * gi + pxl + 1], f1);
                    }
                    const float y0 = fmaf(alpha, f0, bia), y1 = fmaf(alpha, f1, bia);
                    float c0, c1;
                    if (fast) {
                        const v2f c2 = rq_fast2(rqc, v2f{y0, y1});
                        c0 = c2.x; c1 = c2.y;
                    } else {
                        bool b0 = false, b1 = false;
                        c0 = rq_value(rqc, y0, b0);
                        c1 = rq_value(rqc, y1, b1);
                        bad |= img && oc < a.OC && ((b0 && pxl < P) || (b1 && pxl + 1 < P));
                    }
                    if (pxl < P) bp[col * P + pxl] = (uint8_t)(unsigned)c0;
                    if (pxl + 1 < P) bp[col * P + pxl + 1] = (uint8_t)(unsigned)c1;
                }
            }
            if (img && oc0 + 32 <= a.OC) {
                uint8_t *dst = a.rq_out + ((int64_t)(n0 + gi) * a.OC + oc0) * P;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const uint4 d4 = *reinterpret_cast<const uint4 *>(bp + 16 * (64 * k + lane));
                    if (64 * k + lane < 98) *reinterpret_cast<uint4 *>(dst + 16 * (64 * k + lane)) = d4;
                }
            }
        }
        rq_report(a, bad);
    } else {
        // a wave's 32 channels x P pixels of one image are ONE contiguous, 16-byte aligned run of the output
        // ((n OC + oc0) P floats, oc0 % 32 == 0, OC % 4 == 0): the patch is laid out exactly like it and copied flat.
        float *patch = reinterpret_cast<float *>(smem) + wave * (32 * 49);
        const int oc0 = ot * FD_MT + wave * 32;
        const int vrows = min(32, a.OC - oc0);                    // <= 0: nothing to store
        const int nfl = vrows > 0 ? vrows * P : 0;                // floats of the run
        const bool vec_ok = (reinterpret_cast<uintptr_t>(a.out) & 15) == 0;
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int pxl = 32 * tt + (r & 3) + 8 * (r >> 2) + 4 * h;
                    float f = (float)acc[2 * gi + tt][r] + cst;
                    if (need_sx) f = fmaf(-zwp, (float)sxp[64 * gi + pxl], f);
                    if (pxl < P) patch[col * P + pxl] = fmaf(alpha, f, bia);
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            if (n0 + gi < a.N) {
                float *dst = a.out + ((int64_t)(n0 + gi) * a.OC + oc0) * P;
                for (int i = lane; 4 * i < nfl; i += 64) {
                    if (vec_ok && 4 * i + 4 <= nfl) *reinterpret_cast<float4 *>(dst + 4 * i) = *reinterpret_cast<const float4 *>(patch + 4 * i);
                    else for (int b = 4 * i; b < nfl && b < 4 * i + 4; ++b) dst[b] = patch[b];
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// tile variant: 0 = none, 5 / 7 = WIDE with that many column tiles, 8 = SMALL
int flatd_variant(const qe_conv_shape *sh, const qe_qparam *x, const qe_qparam *w)
{
    if (sh->KH != 1 || sh->KW != 1 || sh->stride != 1 || sh->padding != 0) return 0;
    if (x->n_bits != 8 || w->n_bits != 8 || x->n_param != 1) return 0;
    if (sh->IC % FD_CK != 0 || sh->IC < 2 * FD_CK || sh->OC < 1 || sh->N < 1) return 0;
    const int64_t P = (int64_t)sh->H * sh->W;
    if ((int64_t)sh->N * sh->IC * P < 16 || (int64_t)sh->OC * P >= (1ll << 29) || (int64_t)sh->IC * P >= (1ll << 31)) return 0;
    if ((reinterpret_cast<uintptr_t>(w->data) & 15) != 0) return 0;      // weight rows are fetched as aligned 16-byte pieces
    if (P == 49 && sh->OC % 4 == 0 && (reinterpret_cast<uintptr_t>(x->data) & 15) == 0) return 8;
    if ((P % 16 != 0 && P % 16 != 4) || P < 160) return 0;     // a row's last slot holds 16 or 4 valid bytes
    if ((reinterpret_cast<uintptr_t>(x->data) & 3) != 0) return 0;
    auto waste = [&](int t) { return (double)((P + 32 * t - 1) / (32 * t)) * (32 * t) / (double)P; };
    return waste(5) < waste(7) - 0.03 ? 5 : 7;
}

// rq != nullptr: the re-quantising instances (7x7 planes, whole 32-channel strips: OC % 32 == 0, 8-bit codes, one scale)
bool flatd_requant_ok(const qe_conv_shape *sh, const qe_qparam *x, const qe_qparam *w, const RequantHost *rq)
{
    return rq != nullptr && flatd_variant(sh, x, w) == 8 && sh->OC % 32 == 0 && rq->n_bits == 8 && rq->n_param == 1 && rq->out != nullptr &&
           (reinterpret_cast<uintptr_t>(rq->out) & 15) == 0 && !(env_get("QE_FLATD_RQ") && atoi(env_get("QE_FLATD_RQ")) == 0);
}

int launch_flatd(const qe_qparam *x, const qe_qparam *w, const float *bias, const qe_conv_shape *sh, float *out,
                 hipStream_t s, const RequantHost *rq)
{
    const int var = flatd_variant(sh, x, w);
    if (var == 0) return QE_ERR_UNSUPPORTED;
    if (rq != nullptr && !flatd_requant_ok(sh, x, w, rq)) return QE_ERR_UNSUPPORTED;
    FlatdArgs a;
    a.rq_out = nullptr; a.rq_scale = nullptr; a.rq_zero = nullptr; a.rq_status = nullptr;
    a.rq_qmin = a.rq_qmax = a.rq_lo = a.rq_hi = 0.0f; a.rq_offset = 0;
    if (rq != nullptr) {
        a.rq_out = rq->out; a.rq_scale = rq->scale; a.rq_zero = rq->zero;
        a.rq_qmin = rq->qmin; a.rq_qmax = rq->qmax; a.rq_status = rq->status;
        a.rq_offset = rq->sign ? 128u : 0u;                       // tpack.cu:108-111
        a.rq_lo = rq->sign ? -128.0f : 0.0f; a.rq_hi = rq->sign ? 127.0f : 255.0f;
    }
    a.x = static_cast<const uint8_t *>(x->data); a.w = static_cast<const uint8_t *>(w->data);
    a.x_scale = x->scale; a.x_zero = x->zero; a.w_scale = w->scale; a.w_zero = w->zero; a.bias = bias;
    a.x_sign = x->sign; a.w_sign = w->sign; a.w_per_tensor = (w->n_param == 1);
    a.out = out; a.N = sh->N; a.IC = sh->IC; a.OC = sh->OC; a.P = sh->H * sh->W;
    // 8-wave / 256-channel workgroups: measured (profiles/r02l_flatd_w8.txt) -9 % on 512->2048 @7x7, +-3 % on the 14x14
    // layers, +15 % on 2048->512 @7x7 -- halving the activation re-reads does NOT give the -14..-26 % a bytes-through-the-CU
    // model predicts.  On for wide 7x7 layers only; QE_FLATD8=0 | 1 overrides.
    bool w8 = var == 8 && sh->OC >= 1024;
    if (const char *e8 = env_get("QE_FLATD8")) w8 = atoi(e8) != 0 && sh->OC > 128;
    const int MT = w8 ? 256 : 128;
    a.n_oc_tiles = (sh->OC + MT - 1) / MT;
    if (var == 8) { a.tiles_per_image = 1; a.n_pix_tiles = (sh->N + 3) / 4; }
    else { a.tiles_per_image = (a.P + 32 * var - 1) / (32 * var); a.n_pix_tiles = sh->N * a.tiles_per_image; }
    const int64_t per_xcd = ((int64_t)a.n_pix_tiles + 7) / 8;
    a.chunk = (int)(per_xcd < 1 ? 1 : per_xcd);
    if (const char *ci = env_get("QE_CHUNK_IMAGES")) {
        const int64_t k = (int64_t)atoi(ci) * a.tiles_per_image;
        a.chunk = (int)(k < 1 ? 1 : (k < per_xcd ? k : per_xcd));
    }
    const int64_t runs = ((int64_t)a.n_pix_tiles + a.chunk - 1) / a.chunk;
    const int64_t blocks = (runs + 7) / 8 * a.chunk * 8 * a.n_oc_tiles;
    if (blocks > 0x7fffffffLL) return QE_ERR_UNSUPPORTED;
    // ring depth 3 = two stages in flight, two workgroups per CU.  A 6-slot ring (one workgroup per CU, five stages in flight)
    // was 25-60 % slower on every layer (profiles/r02i_flatd_ring.txt): the K loop is not bound by prefetch depth.
#define QE_FD_LAUNCH(NTV, SM, RG, WV)                                                                                      \
    do {                                                                                                                    \
        static const bool ok_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_flatd_kernel<NTV, SM, RG, WV>),   \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, FdGeom<NTV, SM, RG, WV>::LDS) == hipSuccess; \
        (void)ok_;                                                                                                          \
        constexpr size_t lds_ = FdGeom<NTV, SM, RG, WV>::LDS;                                                               \
        hipLaunchKernelGGL((conv_flatd_kernel<NTV, SM, RG, WV>), dim3((unsigned)blocks), dim3(64 * WV), lds_, s, a);        \
    } while (0)
    if (rq != nullptr) {
#define QE_FD_LAUNCH_RQ(WV)                                                                                                 \
    do {                                                                                                                    \
        static const bool ok_ = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_flatd_kernel<8, true, 3, WV, true>), \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, FdGeom<8, true, 3, WV>::LDS) == hipSuccess; \
        (void)ok_;                                                                                                          \
        constexpr size_t lds_ = FdGeom<8, true, 3, WV>::LDS;                                                                \
        hipLaunchKernelGGL((conv_flatd_kernel<8, true, 3, WV, true>), dim3((unsigned)blocks), dim3(64 * WV), lds_, s, a);   \
    } while (0)
        if (w8) QE_FD_LAUNCH_RQ(8); else QE_FD_LAUNCH_RQ(4);
#undef QE_FD_LAUNCH_RQ
    } else if (w8) {
        if (var == 8) QE_FD_LAUNCH(8, true, 3, 8); else if (var == 5) QE_FD_LAUNCH(5, false, 3, 8); else QE_FD_LAUNCH(7, false, 3, 8);
    } else {
        if (var == 8) QE_FD_LAUNCH(8, true, 3, 4); else if (var == 5) QE_FD_LAUNCH(5, false, 3, 4); else QE_FD_LAUNCH(7, false, 3, 4);
    }
#undef QE_FD_LAUNCH
    QE_LAUNCH_CHECK();
    return QE_OK;
}

}  // namespace qe
