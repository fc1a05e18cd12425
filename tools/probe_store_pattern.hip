// probe_store_pattern.hip -- which store shape does a write-bound kernel need to reach the fill rate?
// Writes an [N][OC][P] fp32 tensor tile by tile (128 oc x TP pixels per 256-thread workgroup, wave = 32 oc
// rows), with different lane->address maps per store instruction:
//   mode 0: linear fill (1 KB contiguous per wave instruction), the ceiling
//   mode 1: 8 rows x 128 B   (the flat conv kernel's epilogue)
//   mode 2: 4 rows x 256 B
//   mode 3: 2 rows x 512 B
//   mode 4: 1 row  x 1 KB    (needs TP % 256 == 0)
// NT=1 uses nontemporal stores.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probe_store_pattern tools/probe_store_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float vf4 __attribute__((ext_vector_type(4)));

template <int MODE, bool NT>
__global__ __launch_bounds__(256) void k(float *out, int N, int OC, int P, int TP, int tiles_p, int n_oc_tiles)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bid = blockIdx.x;
    const int grp_sz = 8 * n_oc_tiles;
    const int grp = bid / grp_sz, rem = bid - grp * grp_sz;
    const int pt = grp * 8 + (rem & 7), ot = rem >> 3;
    if (pt >= N * tiles_p) return;
    const int n = pt / tiles_p, p0 = (pt - n * tiles_p) * TP;
    const vf4 v = {1.f, 2.f, 3.f, (float)bid};
    if (MODE == 0) {
        // same bytes per block, but laid out linearly: block b owns bytes [b*128*TP*4, ...)
        float *base = out + ((size_t)(pt * n_oc_tiles + ot) * 128 + wave * 32) * TP;
        for (int i = 0; i < 32 * TP / 256; ++i) {
            vf4 *dst = reinterpret_cast<vf4 *>(base + (size_t)i * 256 + lane * 4);
            if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
        }
        return;
    }
    constexpr int ROWS = MODE == 1 ? 8 : (MODE == 2 ? 4 : (MODE == 3 ? 2 : 1));
    constexpr int LPR = 64 / ROWS;            // lanes per row
    constexpr int WPX = LPR * 4;              // pixels per row per instruction
    const int r = lane / LPR, q = lane % LPR;
    float *ob = out + ((size_t)n * OC + ot * 128 + wave * 32) * P + p0;
    for (int c0 = 0; c0 < TP; c0 += WPX) {
        for (int r0 = 0; r0 < 32; r0 += ROWS) {
            const int px = c0 + 4 * q;
            if (px < TP && p0 + px < P) {
                vf4 *dst = reinterpret_cast<vf4 *>(ob + (size_t)(r0 + r) * P + px);
                if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
            }
        }
    }
}

template <int MODE, bool NT>
static float run(float *out, int N, int OC, int P, int TP)
{
    const int tiles_p = (P + TP - 1) / TP, n_oc = OC / 128;
    const long units = (long)N * tiles_p;
    const long blocks = (units + 7) / 8 * 8 * n_oc;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<MODE, NT>), dim3(blocks), dim3(256), 0, 0, out, N, OC, P, TP, tiles_p, n_oc);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<MODE, NT>), dim3(blocks), dim3(256), 0, 0, out, N, OC, P, TP, tiles_p, n_oc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main()
{
    struct S { int N, OC, P, TP; } shapes[] = {
        {256, 256, 3136, 128}, {256, 256, 3136, 224}, {256, 256, 3136, 256}, {256, 512, 784, 160}, {256, 512, 784, 128},
        {256, 512, 784, 256}, {256, 1024, 196, 224}, {256, 1024, 196, 256}, {256, 2048, 49, 64}};
    for (auto s : shapes) {
        const size_t bytes = (size_t)s.N * s.OC * s.P * 4;
        const int tiles_p = (s.P + s.TP - 1) / s.TP;
        const size_t alloc = (size_t)s.N * tiles_p * s.OC * s.TP * 4 + (1 << 20);
        float *out; hipMalloc(&out, alloc);
        printf("N=%d OC=%d P=%d TP=%d  (%.0f MB)\n", s.N, s.OC, s.P, s.TP, bytes / 1e6);
        float t;
#define R(M, NT, name) t = run<M, NT>(out, s.N, s.OC, s.P, s.TP); printf("   %-22s %.4f ms  %.2f TB/s\n", name, t, (M == 0 ? (double)s.N * tiles_p * s.OC * s.TP * 4 : (double)bytes) / t / 1e9);
        R(0, false, "linear") R(0, true, "linear nt")
        R(1, false, "8 rows x 128 B") R(1, true, "8 rows x 128 B nt")
        R(2, false, "4 rows x 256 B") R(3, false, "2 rows x 512 B") R(3, true, "2 rows x 512 B nt")
        if (s.TP % 256 == 0) { R(4, false, "1 row x 1 KB") }
        hipFree(out);
    }
    return 0;
}
