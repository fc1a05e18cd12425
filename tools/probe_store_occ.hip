// probe_store_occ.hip -- what bounds the conv epilogue's store phase?  Store-only kernels over an [N][OC][P] fp32
// tensor with the flat kernels' tile (128 oc x 224 px, 4 waves x 32 oc, XCD-contiguous block map), varying
//   * resident workgroups per CU (dynamic LDS pads the footprint: 8 / 4 / 3 / 2 per CU),
//   * the instruction shape: 8 rows x 128 B straight from registers, or through the per-wave LDS patch exactly as the
//     conv epilogue does it (4 x ds_write_b128, wait, 4 x ds_read_b128, 4 x global_store_dwordx4 per 32x32 tile),
//   * VGPR pressure irrelevant here (occupancy is set through LDS).
// Launches rotate over enough distinct output buffers (> 1 GB) that no line is still in the 256 MB Infinity Cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float vf4 __attribute__((ext_vector_type(4)));

// flat: every wave writes its 32 rows x NTv pixels as ONE contiguous run when the tile covers whole rows (NTv == P),
// 1 KB per instruction, starting on a 128-byte line
__global__ __launch_bounds__(256) void kflat(float *out, int N, int OC, int P, int NT, int n_oc, long units, int chunk)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bid = blockIdx.x, idx = bid >> 3;
    const int j = idx / n_oc, ot = idx - j * n_oc;
    const int c = j / chunk;
    const long pt = (long)(c * 8 + (bid & 7)) * chunk + (j - c * chunk);
    if (pt >= units) return;
    const int n = (int)pt;                                  // whole-plane tiles only
    float *out_w = out + ((size_t)n * OC + (size_t)ot * 128 + wave * 32) * P;
    const int nf4 = 32 * P / 4;
    for (int i = lane; i < nf4; i += 64) {
        const vf4 o4 = {1.f, 2.f, (float)i, (float)bid};
        *reinterpret_cast<vf4 *>(out_w + 4 * (size_t)i) = o4;
    }
}

// the conv epilogue's form of the flat store for 196-pixel planes: 4 passes of 8 rows; the 16 lanes that own the rows write
// their accumulators into a patch laid out like the output (25 x ds_write_b128), the whole wave copies it out
__global__ __launch_bounds__(256) void kflat_patch(float *out, int N, int OC, int P, int NT, int n_oc, long units, int chunk)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bid = blockIdx.x, idx = bid >> 3;
    const int j = idx / n_oc, ot = idx - j * n_oc;
    const int c = j / chunk;
    const long pt = (long)(c * 8 + (bid & 7)) * chunk + (j - c * chunk);
    if (pt >= units) return;
    const int n = (int)pt;
    float *out_w = out + ((size_t)n * OC + (size_t)ot * 128 + wave * 32) * P;
    float *patch8 = lds + wave * (8 * 196);
    const int col = lane & 31, h = lane >> 5;
    for (int ps = 0; ps < 4; ++ps) {
        if ((col >> 3) == ps) {
            for (int t = 0; t < 7; ++t)
                for (int gq = 0; gq < 4; ++gq) {
                    const int px = 32 * t + 8 * gq + 4 * h;
                    if (px < 196) { const vf4 v = {(float)t, (float)gq, (float)col, (float)bid}; *reinterpret_cast<vf4 *>(patch8 + (col & 7) * 196 + px) = v; }
                }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        float *dst = out_w + (size_t)(8 * ps) * 196;
        for (int i = lane; i < 8 * 49; i += 64) *reinterpret_cast<vf4 *>(dst + 4 * i) = *reinterpret_cast<const vf4 *>(patch8 + 4 * i);
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
}

template <bool PATCH>
__global__ __launch_bounds__(256) void k(float *out, int N, int OC, int P, int NT, int n_oc, long units, int chunk)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bid = blockIdx.x, idx = bid >> 3;
    const int j = idx / n_oc, ot = idx - j * n_oc;
    const int c = j / chunk;
    const long pt = (long)(c * 8 + (bid & 7)) * chunk + (j - c * chunk);
    if (pt >= units) return;
    const int tiles_p = (P + 32 * NT - 1) / (32 * NT);
    const int n = (int)(pt / tiles_p), p0 = (int)(pt - (long)n * tiles_p) * 32 * NT;
    const int NTv = min(32 * NT, P - p0);
    float *out_w = out + ((size_t)n * OC + (size_t)ot * 128 + wave * 32) * P + p0;
    const int rrow = lane >> 3, rq = lane & 7, col = lane & 31, h = lane >> 5;
    float *patch = lds + wave * (32 * 36);
    for (int t = 0; t < NT; ++t) {
        const int q0 = t * 32;
        if (PATCH) {
            for (int gq = 0; gq < 4; ++gq) {
                const vf4 v = {(float)t, (float)gq, (float)col, (float)bid};
                *reinterpret_cast<vf4 *>(patch + col * 36 + 8 * gq + 4 * h) = v;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
        }
        const bool ok = q0 + 4 * rq < NTv;
        for (int i = 0; i < 4; ++i) {
            vf4 o4 = {1.f, 2.f, (float)t, (float)bid};
            if (PATCH) o4 = *reinterpret_cast<const vf4 *>(patch + (8 * i + rrow) * 36 + 4 * rq);
            if (ok) *reinterpret_cast<vf4 *>(out_w + (size_t)(8 * i + rrow) * P + q0 + 4 * rq) = o4;
        }
        if (PATCH) __builtin_amdgcn_s_waitcnt(0xc07f);
    }
}

int main()
{
    struct S { int N, OC, P, NT; } shapes[] = {{256, 1024, 196, 7}, {256, 512, 784, 5}, {256, 256, 3136, 7}, {256, 2048, 49, 2}, {256, 512, 196, 7}, {256, 256, 196, 7}};
    float *bufs[8]; int nb = 0;
    const size_t maxb = (size_t)256 * 256 * 3136 * 4;
    for (; nb < 3; ++nb) if (hipMalloc(&bufs[nb], maxb + (64 << 20)) != hipSuccess) break;
    const int lds_kb[] = {18, 36, 50, 72};           // -> 8, 4, 3, 2 workgroups per CU
    for (auto s : shapes) {
        if (s.P <= 32 * s.NT) {
            const size_t bytes = (size_t)s.N * s.OC * s.P * 4;
            const int per_buf = (int)(maxb / bytes), nsub = per_buf * nb;
            const int n_oc = s.OC / 128; const long units = s.N; const int chunk = (int)((units + 7) / 8);
            const long runs = (units + chunk - 1) / chunk, blocks = (runs + 7) / 8 * chunk * 8 * n_oc;
            hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            int it = 0;
            auto tgt = [&]() { float *p = bufs[(it / per_buf) % nb] + (size_t)(it % per_buf) * (bytes / 4); it = (it + 1) % nsub; return p; };
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kflat, dim3(blocks), dim3(256), 0, 0, tgt(), s.N, s.OC, s.P, s.NT, n_oc, units, chunk);
            (void)hipEventRecord(e0);
            for (int i = 0; i < 12; ++i) hipLaunchKernelGGL(kflat, dim3(blocks), dim3(256), 0, 0, tgt(), s.N, s.OC, s.P, s.NT, n_oc, units, chunk);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("N=%d OC=%d P=%d  flat contiguous runs per wave: %.4f ms  %.2f TB/s\n", s.N, s.OC, s.P, ms / 12, bytes / (ms / 12) / 1e9);
            if (s.P == 196) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kflat_patch), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
                for (int lk : {26, 72}) {
                    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kflat_patch, dim3(blocks), dim3(256), lk * 1024, 0, tgt(), s.N, s.OC, s.P, s.NT, n_oc, units, chunk);
                    (void)hipEventRecord(e0);
                    for (int i = 0; i < 12; ++i) hipLaunchKernelGGL(kflat_patch, dim3(blocks), dim3(256), lk * 1024, 0, tgt(), s.N, s.OC, s.P, s.NT, n_oc, units, chunk);
                    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                    (void)hipEventElapsedTime(&ms, e0, e1);
                    printf("N=%d OC=%d P=%d  flat through 4 x 8-row patches (LDS %d KB/WG): %.4f ms  %.2f TB/s\n", s.N, s.OC, s.P, lk, ms / 12, bytes / (ms / 12) / 1e9);
                }
            }
        }
        const size_t bytes = (size_t)s.N * s.OC * s.P * 4;
        const int per_buf = (int)(maxb / bytes);      // sub-buffers inside each allocation
        const int nsub = per_buf * nb;
        const int tiles_p = (s.P + 32 * s.NT - 1) / (32 * s.NT), n_oc = s.OC / 128;
        const long units = (long)s.N * tiles_p;
        const int chunk = (int)((units + 7) / 8);
        const long runs = (units + chunk - 1) / chunk;
        const long blocks = (runs + 7) / 8 * chunk * 8 * n_oc;
        printf("N=%d OC=%d P=%d NT=%d (%.0f MB, %d rotating targets)\n", s.N, s.OC, s.P, s.NT, bytes / 1e6, nsub);
        for (int lk : lds_kb)
            for (int patch = 0; patch < 2; ++patch) {
                auto fn = patch ? k<true> : k<false>;
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lk * 1024);
                hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
                int it = 0;
                auto tgt = [&]() { float *p = bufs[(it / per_buf) % nb] + (size_t)(it % per_buf) * (bytes / 4); it = (it + 1) % nsub; return p; };
                for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), lk * 1024, 0, tgt(), s.N, s.OC, s.P, s.NT, n_oc, units, chunk);
                (void)hipEventRecord(e0);
                const int reps = 12;
                for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), lk * 1024, 0, tgt(), s.N, s.OC, s.P, s.NT, n_oc, units, chunk);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                printf("  LDS %2d KB/WG  %s: %.4f ms  %.2f TB/s\n", lk, patch ? "via LDS patch" : "from registers", ms / reps, bytes / (ms / reps) / 1e9);
            }
    }
    return 0;
}
