// qe_linear.hip -- quantlinear / quantlinear_float_input for gfx950 (MI355X).
//
// Replaces quantlinear_cuda_kernel (engine/kernels/functions/quantlinear.cu:39-133) and
// quantlinear_float_input_cuda (functions/quantlinear_float_input.cu:36-104): 32x32 shared-memory tiles,
// one output per thread, operands unpacked and dequantised to fp32 inside the K loop.
//
// A Linear is a GEMM whose two operands are both K-contiguous byte rows: exactly the MFMA operand order
// (16 consecutive k per lane), so nothing has to be transposed.  8-bit x 8-bit problems run
//   out[b,o] = bias[o] + sx[b] sw[o] ( S_aw + zw'[o] S_x[b] + zx'[b] S_w[o] + K zx'[b] zw'[o] )
// with a = u ^ 0x80 (signed q, or unsigned q - 128; the 128 goes into z' = z + d), S_aw exact in int32 on
// v_mfma_i32_32x32x32_i8, the row sums S_x / column sums S_w from v_dot4 on the same fragments.  The
// per-ROW activation scale of this op (quantlinear.cu:96) is an epilogue factor like the per-column weight
// scale, so every scale layout is eligible.  Everything else (sub-8-bit operands, fp32 input, K % 16 != 0)
// goes to an order-preserving fp32 kernel that reproduces the reference's k-sequential chain bit for bit
// (with fused multiply-add, as nvcc contracts it).
#include "qe_common.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace qe {

extern unsigned long long *g_mfma_dbg;   // qe_conv_mfma.hip: stamp buffer of -DQE_STAMP diagnostic builds

#ifdef QE_STAMP
__device__ __forceinline__ unsigned long long lin_stamp()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define LIN_ST(i) do { const unsigned long long _t = lin_stamp(); st[i] += _t - tprev; tprev = _t; } while (0)
#else
#define LIN_ST(i) do { } while (0)
#endif

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int lin_unpack(const uint8_t *__restrict__ p, int64_t ele_idx, int n_bits, int sign)
{
    const int64_t bit = ele_idx * n_bits;
    const int64_t byte = bit >> 3;
    const int sh = (int)(bit & 7);
    unsigned v = p[byte] >> sh;
    if (sh + n_bits > 8) v |= (unsigned)p[byte + 1] << (8 - sh);
    v &= (1u << n_bits) - 1u;
    return sign ? (int)v - (1 << (n_bits - 1)) : (int)v;   // value after `-= offset; (T)value` (quantlinear.cu:113,118)
}

struct LinArgs {
    const uint8_t *x;       // packed activations (or nullptr)
    const float *xf;        // fp32 activations (float_input)
    const uint8_t *w;
    const float *x_scale, *x_zero, *w_scale, *w_zero, *bias;
    int x_bits, x_sign, x_per_tensor, w_bits, w_sign, w_per_tensor;
    int64_t B;
    int K, O;
    float *out;
    unsigned long long *dbg;
};

// ---------------------------------------------------------------------------------------------
// Order-preserving fp32 kernel.  32x32 output tile per 256 threads (4 outputs per thread), 32-deep K
// tiles of integer codes (or floats) in LDS; every output runs the reference's chain over k = 0..K-1:
//   packed:      tmp = fmaf((qx + zx) * (qw + zw), sx * sw, tmp)   from 0, + bias last   (quantlinear.cu:113-131)
//   float input: acc = fmaf(x, (qw - zw) * sw, acc)                from 0, + bias last   (float_input.cu:82-102)
// ---------------------------------------------------------------------------------------------
template <bool FLOAT_IN>
__global__ __launch_bounds__(256) void linear_generic_kernel(const LinArgs a)
{
    __shared__ float sA[32][33];
    __shared__ float sB[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // ty 0..7
    const int n_ct = (a.O + 31) / 32;
    const int64_t row0 = (int64_t)(blockIdx.x / n_ct) * 32;
    const int col0 = (int)(blockIdx.x % n_ct) * 32;
    const int col = col0 + tx;
    const int colc = col < a.O ? col : a.O - 1;
    const float zw = a.w_per_tensor ? a.w_zero[0] : a.w_zero[colc];
    const float sw = a.w_per_tensor ? a.w_scale[0] : a.w_scale[colc];
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    float zx[4], s[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t row = row0 + ty + 8 * i;
        const int64_t rc = row < a.B ? row : a.B - 1;
        if constexpr (FLOAT_IN) { zx[i] = 0.0f; s[i] = 0.0f; }
        else {
            zx[i] = a.x_per_tensor ? a.x_zero[0] : a.x_zero[rc];
            s[i] = (a.x_per_tensor ? a.x_scale[0] : a.x_scale[rc]) * sw;      // quantlinear.cu:96
        }
    }
    for (int k0 = 0; k0 < a.K; k0 += 32) {
        // stage: sA[r][k] = activation (row0 + r, k0 + k), sB[c][k] = weight code (col0 + c, k0 + k)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = ty + 8 * i;
            const int64_t row = row0 + r;
            const int k = k0 + tx;
            float va = 0.0f, vb = 0.0f;
            if (row < a.B && k < a.K)
                va = FLOAT_IN ? a.xf[row * a.K + k] : (float)lin_unpack(a.x, row * a.K + k, a.x_bits, a.x_sign);
            const int c = col0 + r;
            if (c < a.O && k < a.K) vb = (float)lin_unpack(a.w, (int64_t)c * a.K + k, a.w_bits, a.w_sign);
            sA[r][tx] = va;
            sB[r][tx] = vb;
        }
        __syncthreads();
        const int kn = min(32, a.K - k0);
        for (int k = 0; k < kn; ++k) {
            const float qw = sB[tx][k];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float xa = sA[ty + 8 * i][k];
                if constexpr (FLOAT_IN) acc[i] = fmaf(xa, (qw - zw) * sw, acc[i]);
                else acc[i] = fmaf((xa + zx[i]) * (qw + zw), s[i], acc[i]);
            }
        }
        __syncthreads();
    }
    const float b = a.bias ? a.bias[colc] : 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t row = row0 + ty + 8 * i;
        if (row < a.B && col < a.O) a.out[row * a.O + col] = acc[i] + b;
    }
}

// ---------------------------------------------------------------------------------------------
// int8 MFMA GEMM, 8-bit x 8-bit, K % 64 == 0.
// Workgroup = 256 threads = 2 x 2 waves; tile = 128 rows (batch) x 256 columns (output features); a wave owns
// 64 x 128 = 2 x 4 MFMA tiles (128 accumulator registers): per 32-deep k-step it reads 2 A + 4 B fragments
// from LDS for 8 MFMAs (24 B/clk/wave: under the CU's 128 B/clk, which a 64 x 64 wave tile would hit exactly).
// Stage = 64 k of both operand tiles (8 + 16 KB), brought in by LDS-DMA (global_load_lds_dwordx4: no staging
// registers) into a ring of 3 buffers, two stages ahead of the MFMAs: a register-staged single-stage prefetch
// left the kernel latency-bound (12 stages x one HBM round trip each: 0.075 ms for 50432x768->768).  The DMA
// writes lane-linear 16-byte pieces, so rows cannot be padded; bank conflicts are avoided by an XOR swizzle
// applied on the SOURCE side (slot s of row r holds k-piece s ^ ((r >> 2) & 3)): the 16 rows of a ds_read_b128
// phase then cover 16 distinct bank quads.  The u ^ 0x80 recode happens on the fragments.  One raw s_barrier per
// stage (s_waitcnt vmcnt(pieces per stage) before it leaves the newest stage in flight; __syncthreads() would drain it).
// D has the batch row on the register and the output feature on the lane: every store instruction writes two
// full 128-byte lines of out[b][:].
// ---------------------------------------------------------------------------------------------
constexpr int LM = 128, LK = 64;
constexpr int L_PA = LM * (LK / 16) / 256;   // A pieces per thread per stage (2)
constexpr int L_RING = 3;

// NJ = column tiles per wave: 4 (tile 128 x 256, 2 workgroups per CU) or 2 (tile 128 x 128, 3 per CU).
template <int NJ>
__global__ __launch_bounds__(256, 2) void linear_mfma_kernel(const LinArgs a)
{
    constexpr int LN = 64 * NJ;
    constexpr int L_PB = LN * (LK / 16) / 256;   // B pieces per thread per stage
    constexpr int L_STAGE = (LM + LN) * LK;      // bytes per ring slot
    extern __shared__ __attribute__((aligned(16))) uint8_t lsm[];   // [L_RING][A 128x64 | B 256x64] + rowc
    float4 *rowc = reinterpret_cast<float4 *>(lsm + L_RING * L_STAGE);   // per batch row: sx, zx', S_x, K zx'
    float4 *colc = rowc + LM;                                            // per output column: sw, zw', bias

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int col = lane & 31, h = lane >> 5;
#ifdef QE_STAMP
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = lin_stamp();
    const unsigned long long tstart = tprev;
#endif
    // column tiles fastest over neighbouring workgroups: they share the activation rows through L2
    const int n_ct = (a.O + LN - 1) / LN;
    const int64_t bid = blockIdx.x;
    const int ct = (int)(bid % n_ct);
    const int64_t rt = bid / n_ct;
    const int64_t m0 = rt * LM;
    const int n0 = ct * LN;

    // epilogue constants go to LDS now, while nothing else is in flight (loaded in the epilogue they cost a
    // serialised global round trip per column tile)
    const float dx = (a.x_sign ? 0.0f : 128.0f), dw = (a.w_sign ? 0.0f : 128.0f);   // a = q - d  ->  z' = z + d
    if (tid < LM) {
        const int64_t row = (m0 + tid < a.B) ? m0 + tid : a.B - 1;
        const float sx = a.x_per_tensor ? a.x_scale[0] : a.x_scale[row];
        const float zxp = (a.x_per_tensor ? a.x_zero[0] : a.x_zero[row]) + dx;
        rowc[tid] = make_float4(sx, zxp, 0.0f, (float)a.K * zxp);
    }
    for (int c = tid; c < LN; c += 256) {
        const int cc = (n0 + c < a.O) ? n0 + c : a.O - 1;
        colc[c] = make_float4(a.w_per_tensor ? a.w_scale[0] : a.w_scale[cc],
                              (a.w_per_tensor ? a.w_zero[0] : a.w_zero[cc]) + dw, a.bias ? a.bias[cc] : 0.0f, 0.0f);
    }
    __syncthreads();

    // DMA pieces: LDS slot e = tid + 256 i <-> (row e >> 2, slot e & 3) receives k-piece (slot ^ swz(row)) of that row;
    // rows past the end are clamped (valid memory, results never stored)
    const int slot = tid & 3;
    const uint8_t *pa[L_PA];
    const uint8_t *pb[L_PB];
#pragma unroll
    for (int i = 0; i < L_PA; ++i) {
        const int r = (tid + 256 * i) >> 2;
        const int64_t row = (m0 + r < a.B) ? m0 + r : a.B - 1;
        pa[i] = a.x + row * a.K + 16 * (slot ^ ((r >> 2) & 3));
    }
#pragma unroll
    for (int i = 0; i < L_PB; ++i) {
        const int r = (tid + 256 * i) >> 2;
        const int c = (n0 + r < a.O) ? n0 + r : a.O - 1;
        pb[i] = a.w + (int64_t)c * a.K + 16 * (slot ^ ((r >> 2) & 3));
    }
    auto issue = [&](int stage) __attribute__((always_inline)) {
        uint8_t *buf = lsm + (stage % L_RING) * L_STAGE;
        const int k0 = stage * LK;
#pragma unroll
        for (int i = 0; i < L_PA; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(pa[i] + k0),
                                             (__attribute__((address_space(3))) void *)(buf + (256 * i + 64 * wave) * 16), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < L_PB; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(pb[i] + k0),
                                             (__attribute__((address_space(3))) void *)(buf + LM * LK + (256 * i + 64 * wave) * 16), 16, 0, 0);
    };

    v16i acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
    int sxa[2] = {0, 0};           // this lane's share of S_x of its two row tiles (row = col, k half h)
    int swa[NJ];                   // ... of S_w of its column tiles
#pragma unroll
    for (int j = 0; j < NJ; ++j) swa[j] = 0;

    const int n_stages = a.K / LK;
    const int swz = (col >> 2) & 3;                       // (row >> 2) & 3 of this lane's fragment rows
    const int a_off = (wm * 64 + col) * LK, b_off = LM * LK + (wn * 32 * NJ + col) * LK;
    issue(0);
    if (n_stages > 1) issue(1);
    LIN_ST(0);   // prologue
    for (int s = 0; s < n_stages; ++s) {
        // stage s has landed for this wave once at most the L_PA + L_PB pieces of stage s+1 are still outstanding
        if (s + 1 < n_stages) __builtin_amdgcn_s_waitcnt(0x0f70 | (L_PA + L_PB)); else __builtin_amdgcn_s_waitcnt(0x0f70);
        LIN_ST(1);   // wait for the stage's DMA
        __builtin_amdgcn_s_barrier();                     // ... for every wave; and ring slot (s+2)%3 is free again
        LIN_ST(2);   // barrier
        if (s + 2 < n_stages) issue(s + 2);
        LIN_ST(3);   // DMA issue
        const uint8_t *buf = lsm + (s % L_RING) * L_STAGE;
#pragma unroll
        for (int ks = 0; ks < LK / 32; ++ks) {
            const int po = 16 * ((2 * ks + h) ^ swz);
            v4i fa[2], fb[NJ];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const v4i *>(buf + a_off + i * 32 * LK + po);
#pragma unroll
            for (int j = 0; j < NJ; ++j) fb[j] = *reinterpret_cast<const v4i *>(buf + b_off + j * 32 * LK + po);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    fa[i][q] ^= (int)0x80808080;          // u - 128: signed q, or unsigned q - 128
                    sxa[i] = __builtin_amdgcn_sdot4(fa[i][q], 0x01010101, sxa[i], false);
                }
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    fb[j][q] ^= (int)0x80808080;
                    swa[j] = __builtin_amdgcn_sdot4(fb[j][q], 0x01010101, swa[j], false);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        LIN_ST(4);   // fragment reads + MFMA
    }

    // ---- epilogue ------------------------------------------------------------------------------
    // S_x joins the row constants (D has the row on the register, S_x lives on the lane that owns the row)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int sx_sum = sxa[i] + __shfl_xor(sxa[i], 32);
        if (wn == 0 && h == 0) rowc[wm * 64 + i * 32 + col].z = (float)sx_sum;
    }
    __syncthreads();
    // D has the batch row on the register and the feature on the lane.  Stored as is, every instruction is a
    // dword store of two 128-byte row pieces, and the store phase (128 instructions per lane) was 40 % of a wave's
    // life.  Each wave therefore turns its 32 x 32 tiles through a private LDS patch (the operand ring is free
    // now) and writes 8 rows x 128 contiguous bytes per global_store_dwordx4: 4x fewer store instructions.
    const bool vec4 = (a.O & 3) == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0;
    float *patch = reinterpret_cast<float *>(lsm) + wave * (32 * 36);
    const int rrow = lane >> 3, rq = lane & 7;
    // Two instances, the choice made once per workgroup: a (wave-uniform) bounds test per tile is a branch per tile, and hipcc
    // opens every block behind a branch with s_waitcnt vmcnt(0) in kernels with LDS-DMA in a loop -- a drain of all outstanding
    // stores in front of every tile (found in linear_mfma8_kernel's epilogue, see there).
    auto tiles = [&](auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int cl = wn * 32 * NJ + j * 32 + col;
            const int c = n0 + cl;
            const float4 cc = colc[cl];                       // sw, zw', bias
            const float sws = (float)(swa[j] + __shfl_xor(swa[j], 32));
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float4 rc = rowc[rl];
                    float t = (float)acc[i][j][r];
                    t = fmaf(cc.y, rc.z, t);
                    t = fmaf(rc.y, sws, t);
                    t = fmaf(rc.w, cc.y, t);
                    v[r] = fmaf(rc.x * cc.x, t, cc.z);
                }
                const int64_t row0 = m0 + wm * 64 + i * 32;
                if constexpr (FULL) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * h) * 36 + col] = v[r];
                    const int c4 = n0 + wn * 32 * NJ + j * 32 + 4 * rq;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int rt = 8 * k + rrow;
                        const float4 o4 = *reinterpret_cast<const float4 *>(patch + rt * 36 + 4 * rq);
                        *reinterpret_cast<float4 *>(a.out + (row0 + rt) * a.O + c4) = o4;
                    }
                } else if (vec4) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * h) * 36 + col] = v[r];
                    const int c4 = n0 + wn * 32 * NJ + j * 32 + 4 * rq;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int rt = 8 * k + rrow;
                        const float4 o4 = *reinterpret_cast<const float4 *>(patch + rt * 36 + 4 * rq);
                        const int64_t row = row0 + rt;
                        if (row < a.B && c4 < a.O) *reinterpret_cast<float4 *>(a.out + row * a.O + c4) = o4;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int64_t row = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (row < a.B && c < a.O) a.out[row * a.O + c] = v[r];
                    }
                }
            }
        }
    };
    // (a wave's own LDS operations complete in order: its reads see its writes, a later tile's writes cannot overtake them)
    if (vec4 && m0 + LM <= a.B && n0 + LN <= a.O) tiles(std::true_type{}); else tiles(std::false_type{});
#ifdef QE_STAMP
    LIN_ST(6);   // epilogue issue
    __builtin_amdgcn_s_waitcnt(0x0f70);
    LIN_ST(7);   // store drain
    if (a.dbg != nullptr && lane == 0) {
        unsigned long long *o = a.dbg + ((size_t)blockIdx.x * 4 + wave) * 10;
        for (int i = 0; i < 8; ++i) o[i] = st[i];
        o[8] = tprev - tstart;
        o[9] = tstart;
    }
#endif
}
template <int NJ> constexpr size_t lin_lds_bytes() { return (size_t)L_RING * (LM + 64 * NJ) * LK + (LM + 64 * NJ) * sizeof(float4); }


// ---------------------------------------------------------------------------------------------
// The same GEMM on a 320 x 256 tile per CU: 512 threads = 2 x 4 waves, a wave owns 160 rows x 64 columns = 5 x 2 MFMA tiles
// (160 accumulator registers, 7 fragments per 10 MFMAs).  Why: linear_mfma_kernel<4> runs TWO 128 x 256 workgroups per CU,
// each fetching its own operand tiles -- (128 + 256) bytes per k for 128 x 256 products, twice: at the matrix pipe's full rate
// that is 48 B/clk through a CU's 64 B/clk vector-memory path, before a single output byte is stored (stamps: 120 cycles to
// ISSUE one of its LDS-DMA instructions, the issue time of a stage's six about what its sixteen MFMAs take).  One 320 x 256
// tile moves (320 + 256) bytes per k for 2.5x the products: 28 B/clk at full rate.  320 rows because the row count that
// matters (50,432 tokens of a ViT-B/16 batch) then makes 158 row tiles: 474 tiles of a 768-column layer are 1.85 rounds of
// the 256 CUs (2 rounds, 92 % full) where 128- or 256-row tiles make 2.31 rounds (3 rounds, 77 % full).
// Stage = 128 k of both operand tiles (72 KB), two buffers, the next stage requested right after the barrier that frees its
// buffer (a full stage of MFMAs -- 40 per wave -- ahead of its use).  128 and not 64 deep because a row piece of a stage is
// then one whole 128-byte cache line: with 64-byte pieces every line crosses the L2 -> L1 path twice (its second half a
// stage later, after 36 KB of other lines have passed through the 32 KB L1), and that path, not the matrix pipe, is what
// bounds this kernel (stamps of the 64-deep form: 150 cycles to issue one LDS-DMA instruction, 28 B/clk per CU).
// One raw s_barrier per stage.  The row sums S_x and column sums S_w (v_dot4 on the fragments) are shared out: column quarter
// wn sums k-step wn of every stage, row half wm every other k-step; the exact integer partials meet in LDS in the epilogue.
// ---------------------------------------------------------------------------------------------
constexpr int L8_TN = 256, L8_K = 128;
// WMW = 2: the 8-wave form above (tile 320 x 256, two stage buffers, one workgroup per CU).
// WMW = 1: 4 waves, tile 160 x 256, ONE stage buffer (52 KB), two workgroups per CU: a workgroup waits out its own operand
// fetch, but the other one computes or stores meanwhile -- for the write-bound shapes (K = 768: six stages, then 164 KB of
// output per tile), where the 8-wave form has nothing to run beside its epilogue.
template <int WMW> struct L8Geom {
    static constexpr int THREADS = 256 * WMW, TM = 160 * WMW, RING = WMW == 2 ? 2 : 1;
    static constexpr int STAGE = (TM + L8_TN) * L8_K;
    static constexpr int PA = TM * (L8_K / 16) / THREADS;      // A pieces per thread per stage (5)
    static constexpr int PB = L8_TN * (L8_K / 16) / THREADS;   // B pieces per thread per stage (4 | 8)
    static constexpr size_t LDS = (size_t)RING * STAGE + (TM + L8_TN) * sizeof(float4);
};

template <int WMW>
__global__ __launch_bounds__(256 * WMW, WMW == 2 ? 1 : 2) void linear_mfma8_kernel(const LinArgs a)
{
    using G = L8Geom<WMW>;
    constexpr int L8_TM = G::TM, L8_RING = G::RING, L8_STAGE = G::STAGE, L8_PA = G::PA, L8_PB = G::PB, THR = G::THREADS;
    extern __shared__ __attribute__((aligned(16))) uint8_t lsm[];
    float4 *rowc = reinterpret_cast<float4 *>(lsm + L8_RING * L8_STAGE);   // per batch row: sx, zx', S_x, K zx'
    float4 *colc = rowc + L8_TM;                                           // per output column: sw, zw', bias, S_w

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = WMW == 2 ? (wave & 1) : 0, wn = WMW == 2 ? (wave >> 1) : wave;
    const int col = lane & 31, h = lane >> 5;
#ifdef QE_STAMP
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = lin_stamp();
    const unsigned long long tstart = tprev;
#endif
    const int n_ct = (a.O + L8_TN - 1) / L8_TN;
    const int64_t bid = blockIdx.x;
    const int ct = (int)(bid % n_ct);
    const int64_t rt = bid / n_ct;
    const int64_t m0 = rt * L8_TM;
    const int n0 = ct * L8_TN;

    // DMA pieces: LDS slot e = tid + 512 i <-> (row e >> 3, slot e & 7) receives k-piece (slot ^ ((row >> 1) & 7)) of that row:
    // a wave-level instruction fetches 8 rows x 128 contiguous bytes = 8 whole cache lines.  ds_read_b128 banks are 256 bytes wide
    // (two rows) and served in the 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32): the 8 even and the 8 odd rows of
    // a group have 8 different (row >> 1) & 7, so a group's 16 reads of one k-piece cover all 16 sixteen-byte slots.
    const int slot = tid & 7;
    uint32_t pa[L8_PA], pb[L8_PB];                         // byte offsets from a.x / a.w (host: B K and O K below 2^32)
#pragma unroll
    for (int i = 0; i < L8_PA; ++i) {
        const int r = (tid + THR * i) >> 3;
        const int64_t row = (m0 + r < a.B) ? m0 + r : a.B - 1;
        pa[i] = (uint32_t)(row * a.K + 16 * (slot ^ ((r >> 1) & 7)));
    }
#pragma unroll
    for (int i = 0; i < L8_PB; ++i) {
        const int r = (tid + THR * i) >> 3;
        const int c = (n0 + r < a.O) ? n0 + r : a.O - 1;
        pb[i] = (uint32_t)((int64_t)c * a.K + 16 * (slot ^ ((r >> 1) & 7)));
    }
    // piece q of a stage: 0..4 = A, 5..8 = B.  The requests of stage s + 1 are spread over the k-steps of stage s: issued in one
    // burst behind the barrier, the 72 instructions of a workgroup queue at the CU's one address path and every wave sits in
    // its issue slot (180 cycles per instruction, stamps) with no MFMA behind it -- 1,600 cycles of a 7,300-cycle stage.
    auto issue_piece = [&](int stage, auto q_tag) __attribute__((always_inline)) {
        constexpr int q = decltype(q_tag)::value;
        uint8_t *buf = lsm + (stage % L8_RING) * L8_STAGE;
        const int k0 = stage * L8_K;
        if constexpr (q >= L8_PA + L8_PB) { }
        else if constexpr (q < L8_PA)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.x + (pa[q] + (uint32_t)k0)),
                                             (__attribute__((address_space(3))) void *)(buf + (THR * q + 64 * wave) * 16), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.w + (pb[q - L8_PA] + (uint32_t)k0)),
                                             (__attribute__((address_space(3))) void *)(buf + L8_TM * L8_K + (THR * (q - L8_PA) + 64 * wave) * 16), 16, 0, 0);
    };
#define QE_L8_ISSUE(ST, Q) issue_piece(ST, std::integral_constant<int, Q>{})
    auto issue = [&](int stage) __attribute__((always_inline)) {
        QE_L8_ISSUE(stage, 0); QE_L8_ISSUE(stage, 1); QE_L8_ISSUE(stage, 2); QE_L8_ISSUE(stage, 3); QE_L8_ISSUE(stage, 4);
        QE_L8_ISSUE(stage, 5); QE_L8_ISSUE(stage, 6); QE_L8_ISSUE(stage, 7); QE_L8_ISSUE(stage, 8);
        if constexpr (L8_PA + L8_PB > 9) { QE_L8_ISSUE(stage, 9); QE_L8_ISSUE(stage, 10); QE_L8_ISSUE(stage, 11); QE_L8_ISSUE(stage, 12); }
    };
    const int n_stages = a.K / L8_K;
    issue(0);                                             // in flight while the epilogue constants are fetched

    const float dx = (a.x_sign ? 0.0f : 128.0f), dw = (a.w_sign ? 0.0f : 128.0f);   // a = q - d  ->  z' = z + d
    if (tid < L8_TM) {
        const int64_t row = (m0 + tid < a.B) ? m0 + tid : a.B - 1;
        const float sx = a.x_per_tensor ? a.x_scale[0] : a.x_scale[row];
        const float zxp = (a.x_per_tensor ? a.x_zero[0] : a.x_zero[row]) + dx;
        rowc[tid] = make_float4(sx, zxp, 0.0f, (float)a.K * zxp);
    }
    if (tid < L8_TN) {
        const int cc = (n0 + tid < a.O) ? n0 + tid : a.O - 1;
        colc[tid] = make_float4(a.w_per_tensor ? a.w_scale[0] : a.w_scale[cc],
                                (a.w_per_tensor ? a.w_zero[0] : a.w_zero[cc]) + dw, a.bias ? a.bias[cc] : 0.0f, 0.0f);
    }

    v16i acc[5][2];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
    int sxa[5] = {0, 0, 0, 0, 0};  // this lane's share of S_x of its five row tiles (row = col, k half h) over ITS k-steps (ks == wn)
    int swa[2] = {0, 0};           // ... of S_w of its two column tiles over its k-steps (ks & 1 == wm)

    const int swz = (col >> 1) & 7;                       // (row >> 1) & 7 of this lane's fragment rows (row = 32 t + col)
    const int a_off = (wm * 160 + col) * L8_K, b_off = L8_TM * L8_K + (wn * 64 + col) * L8_K;
    LIN_ST(0);   // prologue
    for (int s = 0; s < n_stages; ++s) {
        __builtin_amdgcn_s_waitcnt(0x0f70);               // vmcnt(0): this wave's pieces of stage s (the only ones in flight) landed
        LIN_ST(1);   // wait for the stage's DMA
        __builtin_amdgcn_s_barrier();                     // ... every wave's; and every wave is done with stage s - 1: its buffer is free
        LIN_ST(2);   // barrier
        const bool more = L8_RING == 2 && s + 1 < n_stages;   // one buffer: the next stage is requested behind the barrier at the loop's end
        LIN_ST(3);
        const uint8_t *buf = lsm + (s % L8_RING) * L8_STAGE;
        // k-step pipeline of ONE wave: the B fragments of k-step ks + 1 are requested while the MFMAs of k-step ks run; the A
        // fragments are waited for one by one (counted lgkmcnt), each recoded right in front of its two MFMAs, so their LDS
        // latency and the recoding VALU work sit under the MFMAs of the rows before them (sched_group_barrier pins that order:
        // left alone, hipcc hoists all 28 v_xor in front of the first MFMA behind one lgkmcnt(0)).
        v4i fb[2], fbn[2];
        {
            const int po0 = 16 * (h ^ swz);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const v4i *>(buf + b_off + j * 32 * L8_K + po0);
        }
#pragma unroll
        for (int ks = 0; ks < L8_K / 32; ++ks) {
            if (more) {
                if (ks == 0) { QE_L8_ISSUE(s + 1, 0); QE_L8_ISSUE(s + 1, 1); QE_L8_ISSUE(s + 1, 2); }
                if (ks == 1) { QE_L8_ISSUE(s + 1, 3); QE_L8_ISSUE(s + 1, 4); }
                if (ks == 2) { QE_L8_ISSUE(s + 1, 5); QE_L8_ISSUE(s + 1, 6); }
                if (ks == 3) { QE_L8_ISSUE(s + 1, 7); QE_L8_ISSUE(s + 1, 8); }
            }
            __builtin_amdgcn_sched_barrier(0);
            const int po = 16 * ((2 * ks + h) ^ swz);
            v4i fa[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) fa[i] = *reinterpret_cast<const v4i *>(buf + a_off + i * 32 * L8_K + po);
            if (ks + 1 < L8_K / 32) {
                const int pon = 16 * ((2 * (ks + 1) + h) ^ swz);
#pragma unroll
                for (int j = 0; j < 2; ++j) fbn[j] = *reinterpret_cast<const v4i *>(buf + b_off + j * 32 * L8_K + pon);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) fb[j][q] ^= (int)0x80808080;
            if (WMW == 1 || (ks & 1) == wm) {             // S_w: the two row halves take alternate k-steps (wave-uniform)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) swa[j] = __builtin_amdgcn_sdot4(fb[j][q], 0x01010101, swa[j], false);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 5; ++i) {
#pragma unroll
                for (int q = 0; q < 4; ++q) fa[i][q] ^= (int)0x80808080;          // u - 128: signed q, or unsigned q - 128
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // 4 VALU (recode of row tile i)
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // 2 MFMA
            }
            __builtin_amdgcn_sched_barrier(0);
            if (ks == wn) {                               // S_x: column quarter wn takes k-step wn of every stage (wave-uniform)
#pragma unroll
                for (int i = 0; i < 5; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) sxa[i] = __builtin_amdgcn_sdot4(fa[i][q], 0x01010101, sxa[i], false);
            }
            if (ks + 1 < L8_K / 32) { fb[0] = fbn[0]; fb[1] = fbn[1]; }
        }
        LIN_ST(4);   // fragment reads + MFMA
        if constexpr (L8_RING == 1) {
            if (s + 1 < n_stages) {
                __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0): this wave's fragment reads are done
                __builtin_amdgcn_s_barrier();             // every wave's: the one buffer is free
                issue(s + 1);
            }
        }
    }

    // ---- epilogue ------------------------------------------------------------------------------
    __syncthreads();                                      // every wave is done with the operand ring; the constants are in place
    // partial sums (exact integers) meet in LDS: S_x from the four column quarters, S_w from the two row halves
    int *part = reinterpret_cast<int *>(lsm);             // [4][320] S_x partials, then [2][256] S_w partials
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int sx_sum = sxa[i] + __shfl_xor(sxa[i], 32);
        if (h == 0) part[wn * L8_TM + wm * 160 + i * 32 + col] = sx_sum;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int sw_sum = swa[j] + __shfl_xor(swa[j], 32);
        if (h == 0) part[4 * L8_TM + wm * L8_TN + wn * 64 + j * 32 + col] = sw_sum;
    }
    __syncthreads();
    if (tid < L8_TM) rowc[tid].z = (float)(part[tid] + part[L8_TM + tid] + part[2 * L8_TM + tid] + part[3 * L8_TM + tid]);
    if (tid < L8_TN) colc[tid].w = (float)(part[4 * L8_TM + tid] + (WMW == 2 ? part[4 * L8_TM + L8_TN + tid] : 0));
    __syncthreads();                                      // sums visible
    // A wave's own LDS operations complete in order, so its reads see its writes and a later tile's writes cannot overtake
    // the reads of the tile before: no lgkmcnt drain between them.
    // symmetric operands (zx' = 0 for the wave's 160 rows, zw' = 0 for its 64 columns): out = bias + (sx sw) S_aw -- the three
    // correction terms of the general form are exact zeros there, so both forms give the same bits; no per-row constants read
    bool sym;
    {
        bool nz = false;
#pragma unroll
        for (int q = 0; q < 3; ++q) { const int rr = lane + 64 * q; if (rr < 160) nz |= rowc[wm * 160 + rr].y != 0.0f; }
        nz |= colc[wn * 64 + lane].y != 0.0f;
        sym = __builtin_amdgcn_ballot_w64(nz) == 0ull;
    }
    const float sx_all = a.x_scale[0];
    const bool fast = sym && a.x_per_tensor;
    // The epilogue is instantiated twice and the choice made ONCE: inside it there is no branch.  hipcc's wait-count pass
    // starts every basic block that follows a branch with s_waitcnt vmcnt(0) while the kernel holds LDS-DMA instructions it
    // cannot prove retired -- with a (wave-uniform) bounds test around every tile's stores that was a drain of ALL
    // outstanding global stores in front of every tile: 21 k cycles of epilogue per wave whatever the store shape (stamps).
    auto convert = [&](int i, int j, float (&v)[16], auto fast_tag) __attribute__((always_inline)) {
        const float4 cc = colc[wn * 64 + j * 32 + col];   // sw, zw', bias, S_w
        if constexpr (decltype(fast_tag)::value) {
            const float al = sx_all * cc.x;
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = fmaf(al, (float)acc[i][j][r], cc.z);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float4 rc = rowc[wm * 160 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
                float t = (float)acc[i][j][r];
                t = fmaf(cc.y, rc.z, t);
                t = fmaf(rc.y, cc.w, t);
                t = fmaf(rc.w, cc.y, t);
                v[r] = fmaf(rc.x * cc.x, t, cc.z);
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // at most four rows' constants live (else 64 VGPRs of them: scratch)
            }
        }
    };
    // a patch is the wave's 32 rows x 64 columns of one row tile (both column tiles): read back as 4 rows x 256 contiguous
    // bytes per store instruction (16 lanes per row)
    float *patch0 = reinterpret_cast<float *>(lsm) + wave * (32 * 64);
    const int r4 = lane >> 4, q16 = lane & 15;
    const uint32_t voff = (uint32_t)r4 * (uint32_t)a.O + 4u * (uint32_t)q16;   // elements; r4 O + 64 < 2^31 (O < 2^29)
    auto tiles = [&](auto fast_tag, auto full_tag) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float v[16];
                convert(i, j, v, fast_tag);
#pragma unroll
                for (int r = 0; r < 16; ++r) patch0[((r & 3) + 8 * (r >> 2) + 4 * h) * 64 + j * 32 + col] = v[r];
            }
            const int64_t row0 = m0 + wm * 160 + i * 32;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float4 o4 = *reinterpret_cast<const float4 *>(patch0 + (4 * k + r4) * 64 + 4 * q16);
                // wave-uniform base (scalar registers) + ONE per-lane 32-bit offset for all 40 stores of the wave
                float *base = a.out + (row0 + 4 * k) * (int64_t)a.O + (n0 + wn * 64);
                if constexpr (decltype(full_tag)::value) *reinterpret_cast<float4 *>(base + voff) = o4;
                else if (row0 + 4 * k + r4 < a.B) *reinterpret_cast<float4 *>(base + voff) = o4;
            }
        }
    };
    // O % 256 == 0 and `out` 16-byte aligned (host); only the LAST row tile can be ragged: it alone takes the form with a row test
    if (m0 + L8_TM <= a.B) { if (fast) tiles(std::true_type{}, std::true_type{}); else tiles(std::false_type{}, std::true_type{}); }
    else tiles(std::false_type{}, std::false_type{});
#ifdef QE_STAMP
    LIN_ST(6);   // epilogue issue
    __builtin_amdgcn_s_waitcnt(0x0f70);
    LIN_ST(7);   // store drain
    if (a.dbg != nullptr && lane == 0) {
        unsigned long long *o = a.dbg + ((size_t)blockIdx.x * (4 * WMW) + wave) * 10;
        for (int i = 0; i < 8; ++i) o[i] = st[i];
        o[8] = tprev - tstart;
        o[9] = tstart;
    }
#endif
}


// ---------------------------------------------------------------------------------------------
// quantlinear_float_input on the matrix cores (round 3): fp32 activations x 8-bit weight codes, K % 32 == 0.
//   out[b,o] = bias[o] + sw[o] ( S_xq[b,o] - zw[o] S_x[b] ),   S_xq = sum_k x q_w,   S_x = sum_k x
// S_xq runs on v_mfma_f32_32x32x16_bf16 with the EXACT three-way split of the activations the float-input convolution
// uses (qe_conv_f32.hip): x = x1 + x2 + x3, each part's significand <= 8 bits (bf16), every product with an integer code
// |q| <= 255 (bf16-exact) exact in fp32; only the fp32 accumulation rounds.  (A non-finite x gives NaN in the remainder
// terms where the reference's chain keeps +-inf: declared divergence, as for the convolution.)
// Workgroup = 2 x 2 waves, tile 128 x 128, wave 64 x 64 = 2 x 2 MFMA tiles x 3 splits; stage = 32 k: the thread that
// fetched 4 consecutive k of a row splits them and writes 3 x 8 bytes; weights: 16 codes -> 16 bf16 per thread.
// LDS images [row][32 k] bf16 (64-byte rows), 16-byte pieces XOR-swizzled as in linear_mfma_kernel.
// ---------------------------------------------------------------------------------------------
typedef __bf16 v8bf_l __attribute__((ext_vector_type(8)));
typedef float v16f_l __attribute__((ext_vector_type(16)));
constexpr int LF_T = 128, LF_K = 32;
constexpr int LF_PLANE = LF_T * LF_K * 2;                 // bytes of one [128][32] bf16 image (8 KB)
constexpr size_t linf_lds_bytes() { return (size_t)4 * LF_PLANE + LF_T * sizeof(float) + 2 * LF_T * sizeof(float4); }

__global__ __launch_bounds__(256, 2) void linear_f32_mfma_kernel(const LinArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lsm[];   // 3 activation split images, 1 weight image, row sums, column constants
    float *rowsum = reinterpret_cast<float *>(lsm + 4 * LF_PLANE);
    float4 *colc = reinterpret_cast<float4 *>(rowsum + LF_T);        // sw, zw, bias, 0

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int col = lane & 31, h = lane >> 5;
    const int n_ct = (a.O + LF_T - 1) / LF_T;
    const int64_t bid = blockIdx.x;
    const int ct = (int)(bid % n_ct);
    const int64_t m0 = (bid / n_ct) * LF_T;
    const int n0 = ct * LF_T;

    if (tid < LF_T) {
        const int cc = (n0 + tid < a.O) ? n0 + tid : a.O - 1;
        colc[tid] = make_float4(a.w_per_tensor ? a.w_scale[0] : a.w_scale[cc], a.w_per_tensor ? a.w_zero[0] : a.w_zero[cc],
                                a.bias ? a.bias[cc] : 0.0f, 0.0f);
    }

    // activations: thread <-> (row r = (tid >> 3) + 32 i, k quad q = tid & 7): one float4 per pass i
    const int q8 = tid & 7;
    const float *xrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (tid >> 3) + 32 * i;
        const int64_t row = (m0 + r < a.B) ? m0 + r : a.B - 1;
        xrow[i] = a.xf + row * a.K + 4 * q8;
    }
    // weights: thread <-> (column c = tid >> 1, 16-code half hh = tid & 1)
    const int wc = tid >> 1, whh = tid & 1;
    const uint8_t *wrow = a.w + (int64_t)((n0 + wc < a.O) ? n0 + wc : a.O - 1) * a.K + 16 * whh;
    const float wshift = a.w_sign ? 128.0f : 0.0f;       // stored u -> q = u - 128 (signed) | u

    float4 xv[4];
    uint4 wv;
    auto fetch = [&](int st) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) xv[i] = *reinterpret_cast<const float4 *>(xrow[i] + st * LF_K);
        wv = *reinterpret_cast<const uint4 *>(wrow + st * LF_K);
    };
    float sx[4] = {0.0f, 0.0f, 0.0f, 0.0f};              // this thread's share of S_x of its four rows
    auto stage = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (tid >> 3) + 32 * i;
            float x[4] = {xv[i].x, xv[i].y, xv[i].z, xv[i].w};
            sx[i] += (x[0] + x[1]) + (x[2] + x[3]);
            // piece (16 bytes = 8 k) q8 >> 1 of the row, its half q8 & 1
            uint8_t *dst = lsm + r * 64 + 16 * ((q8 >> 1) ^ ((r >> 2) & 3)) + 8 * (q8 & 1);
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) {
                uint32_t u[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    u[j] = __float_as_uint(x[j]) & 0xffff0000u;           // sp == 2: the remainder has <= 8 significant bits
                    x[j] -= __uint_as_float(u[j]);
                }
                *reinterpret_cast<uint2 *>(dst + sp * LF_PLANE) =
                    make_uint2(__builtin_amdgcn_perm(u[1], u[0], 0x07060302u), __builtin_amdgcn_perm(u[3], u[2], 0x07060302u));
            }
        }
        // 16 codes -> 16 bf16 (two 16-byte pieces: k 16 hh .. 16 hh + 7, + 8 .. + 15)
        const uint32_t ww[4] = {wv.x, wv.y, wv.z, wv.w};
        uint32_t pk[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t b0 = (ww[j >> 1] >> (16 * (j & 1))) & 0xffu, b1 = (ww[j >> 1] >> (16 * (j & 1) + 8)) & 0xffu;
            const float f0 = (float)b0 - wshift, f1 = (float)b1 - wshift;   // |q| <= 255: exact in bf16
            pk[j] = __builtin_amdgcn_perm(__float_as_uint(f1), __float_as_uint(f0), 0x07060302u);
        }
        uint8_t *wd = lsm + 3 * LF_PLANE + wc * 64;
        const int sw3 = (wc >> 2) & 3;
        *reinterpret_cast<uint4 *>(wd + 16 * ((2 * whh) ^ sw3)) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        *reinterpret_cast<uint4 *>(wd + 16 * ((2 * whh + 1) ^ sw3)) = make_uint4(pk[4], pk[5], pk[6], pk[7]);
    };

    v16f_l acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int n_stages = a.K / LF_K;
    const int swz = (col >> 2) & 3;
    const int a_off = (wm * 64 + col) * 64, b_off = 3 * LF_PLANE + (wn * 64 + col) * 64;
    fetch(0);
    for (int st = 0; st < n_stages; ++st) {
        stage();
        __syncthreads();
        if (st + 1 < n_stages) fetch(st + 1);             // in flight under the MFMAs
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int po = 16 * ((2 * ks + h) ^ swz);
            v8bf_l fb[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const v8bf_l *>(lsm + b_off + j * 32 * 64 + po);
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const v8bf_l fa = *reinterpret_cast<const v8bf_l *>(lsm + sp * LF_PLANE + a_off + i * 32 * 64 + po);
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb[j], acc[i][j], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    // S_x: the 8 threads of a row (consecutive lanes) add their shares
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v = sx[i];
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
        if (q8 == 0) rowsum[(tid >> 3) + 32 * i] = v;
    }
    __syncthreads();
    // D: batch row on the register, feature on the lane
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int cl = wn * 64 + j * 32 + col;
        const int c = n0 + cl;
        const float4 cc = colc[cl];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int64_t row = m0 + rl;
                const float v = fmaf(cc.x, fmaf(-cc.y, rowsum[rl], acc[i][j][r]), cc.z);
                if (row < a.B && c < a.O) a.out[row * a.O + c] = v;
            }
        }
    }
}

static bool linf_mfma_eligible(const float *x, const qe_qparam *w, int64_t B, int K, int O)
{
    if (const char *e = env_get("QE_LIN_F32_MFMA")) { if (atoi(e) == 0) return false; }
    return w->n_bits == 8 && (K % LF_K) == 0 && K >= LF_K && B > 0 && O > 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
           (reinterpret_cast<uintptr_t>(w->data) & 15) == 0;
}

static int check_lin_q(const qe_qparam *q, int64_t n_expected)
{
    if (q == nullptr || q->data == nullptr || q->scale == nullptr || q->zero == nullptr) return QE_ERR_ARG;
    if (!(q->n_bits > 0 && q->n_bits <= 8)) return QE_ERR_NBITS;
    if (q->n_param != 1 && q->n_param != n_expected) return QE_ERR_ARG;
    return QE_OK;
}

static bool lin_mfma_eligible(const qe_qparam *x, const qe_qparam *w, int64_t B, int K, int O)
{
    // K < 2^17: int32 accumulation of products up to 2^14 cannot overflow (deeper reductions take the fp32 kernel, as the reference's fp32 sum does)
    return x->n_bits == 8 && w->n_bits == 8 && (K % LK) == 0 && K >= LK && K < (1 << 17) && B > 0 && O > 0 &&
           (reinterpret_cast<uintptr_t>(x->data) & 15) == 0 && (reinterpret_cast<uintptr_t>(w->data) & 15) == 0;
}

}  // namespace qe

extern "C" int qe_quantlinear_path(const qe_qparam *x, const qe_qparam *w, int64_t B, int32_t K, int32_t O)
{
    if (x == nullptr || w == nullptr) return 0;
    return qe::lin_mfma_eligible(x, w, B, K, O) ? 1 : 0;
}

extern "C" int qe_quantlinear(const qe_qparam *x, const qe_qparam *w, const float *bias,
                              int64_t B, int32_t K, int32_t O, float *out, qe_stream_t stream)
{
    using namespace qe;
    if (B < 0 || K < 0 || O < 0) return QE_ERR_ARG;
    int rc;
    if ((rc = check_lin_q(x, B)) != QE_OK) return rc;
    if ((rc = check_lin_q(w, O)) != QE_OK) return rc;
    if (out == nullptr && B * O > 0) return QE_ERR_ARG;
    if (B == 0 || O == 0) return QE_OK;
    LinArgs a;
    a.x = static_cast<const uint8_t *>(x->data); a.xf = nullptr; a.w = static_cast<const uint8_t *>(w->data);
    a.x_scale = x->scale; a.x_zero = x->zero; a.w_scale = w->scale; a.w_zero = w->zero; a.bias = bias;
    a.x_bits = x->n_bits; a.x_sign = x->sign; a.x_per_tensor = x->n_param == 1;
    a.w_bits = w->n_bits; a.w_sign = w->sign; a.w_per_tensor = w->n_param == 1;
    a.B = B; a.K = K; a.O = O; a.out = out; a.dbg = g_mfma_dbg;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (lin_mfma_eligible(x, w, B, K, O)) {
        // tile width: 256 columns unless that leaves the chip under-filled; QE_LIN_NJ=2|4 overrides (tuning)
        int nj = 4;
        if (((B + LM - 1) / LM) * ((O + 255) / 256) < kNumCU) nj = 2;   // under-filled chip (the ViT head: 8 workgroups): twice as many, half as wide (17.8 -> 11.5 us)
        if (const char *e = env_get("QE_LIN_NJ")) nj = atoi(e) == 2 ? 2 : 4;
        const int ln = 64 * nj;
        const int64_t blocks = ((B + LM - 1) / LM) * ((O + ln - 1) / ln);
        if (blocks > 0x7fffffffLL) return QE_ERR_UNSUPPORTED;
        // more than 64 KB of dynamic LDS needs the attribute once
        static const bool raised =
            hipFuncSetAttribute(reinterpret_cast<const void *>(&linear_mfma_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lin_lds_bytes<4>()) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void *>(&linear_mfma_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lin_lds_bytes<2>()) == hipSuccess;
        (void)raised;
        // 128-deep stages: 320 x 256 tiles (one 8-wave workgroup per CU) when the reduction is deep, 160 x 256 tiles (two 4-wave
        // workgroups per CU) when the layer is bound by its stores (K <= 1024) -- either when the problem fills the chip with
        // them (O % 256 == 0: whole column tiles).  QE_LIN8=0: never, 1: the 8-wave form, 2: the 4-wave form
        int big = 0;
        if ((K % L8_K) == 0 && (O % L8_TN) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && B * (int64_t)K < (1ll << 32) &&
            (int64_t)O * K < (1ll << 32) && O < (1 << 28)) {
            // thresholds from tools/bench_linear.py at 256 / 64 / 16 images (profiles/r03zz_lin_small_batches.txt): the big tiles
            // still win at 12,608 rows (120 / 237 tiles), the 64-deep kernel's smaller tiles at 3,152 rows unless O is wide
            if (K > 1024 && (B / 320) * (O / L8_TN) >= kNumCU / 3) big = 1;
            else if (K <= 1024 && (B / 160) * (O / L8_TN) >= kNumCU / 2) big = 2;
            if (const char *e = env_get("QE_LIN8")) big = atoi(e);
            if (big < 0 || big > 2) big = 0;
        }
        if (big != 0 && !env_get("QE_LIN_NJ")) {
            static const bool raised8 =
                hipFuncSetAttribute(reinterpret_cast<const void *>(&linear_mfma8_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)L8Geom<2>::LDS) == hipSuccess &&
                hipFuncSetAttribute(reinterpret_cast<const void *>(&linear_mfma8_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)L8Geom<1>::LDS) == hipSuccess;
            (void)raised8;
            const int tm = big == 1 ? 320 : 160;
            const int64_t blocks8 = ((B + tm - 1) / tm) * (O / L8_TN);
            if (blocks8 > 0x7fffffffLL) return QE_ERR_UNSUPPORTED;
            if (big == 1) hipLaunchKernelGGL(linear_mfma8_kernel<2>, dim3((unsigned)blocks8), dim3(512), L8Geom<2>::LDS, s, a);
            else          hipLaunchKernelGGL(linear_mfma8_kernel<1>, dim3((unsigned)blocks8), dim3(256), L8Geom<1>::LDS, s, a);
        } else
        if (nj == 4) hipLaunchKernelGGL(linear_mfma_kernel<4>, dim3((unsigned)blocks), dim3(256), lin_lds_bytes<4>(), s, a);
        else         hipLaunchKernelGGL(linear_mfma_kernel<2>, dim3((unsigned)blocks), dim3(256), lin_lds_bytes<2>(), s, a);
    } else {
        const int64_t blocks = ((B + 31) / 32) * ((O + 31) / 32);
        if (blocks > 0x7fffffffLL) return QE_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(linear_generic_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, a);
    }
    QE_LAUNCH_CHECK();
    return QE_OK;
}

extern "C" int qe_quantlinear_float_input(const float *x, const qe_qparam *w, const float *bias,
                                          int64_t B, int32_t K, int32_t O, float *out, qe_stream_t stream)
{
    using namespace qe;
    if (B < 0 || K < 0 || O < 0) return QE_ERR_ARG;
    int rc;
    if ((rc = check_lin_q(w, O)) != QE_OK) return rc;
    if ((x == nullptr && B * K > 0) || (out == nullptr && B * O > 0)) return QE_ERR_ARG;
    if (B == 0 || O == 0) return QE_OK;
    LinArgs a;
    a.x = nullptr; a.xf = x; a.w = static_cast<const uint8_t *>(w->data);
    a.x_scale = nullptr; a.x_zero = nullptr; a.w_scale = w->scale; a.w_zero = w->zero; a.bias = bias;
    a.x_bits = 0; a.x_sign = 0; a.x_per_tensor = 1;
    a.w_bits = w->n_bits; a.w_sign = w->sign; a.w_per_tensor = w->n_param == 1;
    a.B = B; a.K = K; a.O = O; a.out = out; a.dbg = nullptr;
    if (linf_mfma_eligible(x, w, B, K, O)) {
        const int64_t blocks_m = ((B + LF_T - 1) / LF_T) * ((O + LF_T - 1) / LF_T);
        if (blocks_m > 0x7fffffffLL) return QE_ERR_UNSUPPORTED;
        static const bool raised_f = hipFuncSetAttribute(reinterpret_cast<const void *>(&linear_f32_mfma_kernel),
                                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)linf_lds_bytes()) == hipSuccess;
        (void)raised_f;
        hipLaunchKernelGGL(linear_f32_mfma_kernel, dim3((unsigned)blocks_m), dim3(256), linf_lds_bytes(), static_cast<hipStream_t>(stream), a);
        QE_LAUNCH_CHECK();
        return QE_OK;
    }
    const int64_t blocks = ((B + 31) / 32) * ((O + 31) / 32);
    if (blocks > 0x7fffffffLL) return QE_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(linear_generic_kernel<true>, dim3((unsigned)blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), a);
    QE_LAUNCH_CHECK();
    return QE_OK;
}

extern "C" int qe_quantlinear_float_input_path(const float *x, const qe_qparam *w, int64_t B, int32_t K, int32_t O)
{
    if (w == nullptr) return 0;
    return qe::linf_mfma_eligible(x, w, B, K, O) ? 1 : 0;
}
