// probe_store_pattern2.hip -- store-only kernels over an [N][OC][P] fp32 tensor: how do tile shape and the
// block -> tile map change the achieved write rate?  (follow-up of probe_store_pattern.hip)
// block = OCB oc rows x TP pixels, 4 waves split the rows; one instruction = ROWS rows x (256/ROWS) pixels.
// map 0: groups of 8 pixel tiles x all oc tiles (the conv kernels' XCD-aware map)
// map 1: oc tile fastest      map 2: pixel tile fastest (all of oc tile 0 first)
// map 3: each XCD (bid & 7) owns a contiguous eighth of the units, oc tile fastest inside
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float vf4 __attribute__((ext_vector_type(4)));

template <int ROWS>
__global__ __launch_bounds__(256) void k(float *out, int N, int OC, int P, int TP, int OCB, int tiles_p, int n_oc, int map, long units)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long bid = blockIdx.x;
    long pt; int ot;
    if (map == 0) {
        const int grp_sz = 8 * n_oc;
        const long grp = bid / grp_sz; const int rem = (int)(bid - grp * grp_sz);
        pt = grp * 8 + (rem & 7); ot = rem >> 3;
    } else if (map == 1) { ot = (int)(bid % n_oc); pt = bid / n_oc; }
    else if (map == 2) { pt = bid % units; ot = (int)(bid / units); }
    else {
        const long total = units * n_oc, per = (total + 7) / 8;
        const long u = (bid & 7) * per + (bid >> 3);
        if ((bid >> 3) >= per || u >= total) return;
        ot = (int)(u % n_oc); pt = u / n_oc;
    }
    if (pt >= units) return;
    const int n = (int)(pt / tiles_p), p0 = (int)(pt - (long)n * tiles_p) * TP;
    const vf4 v = {1.f, 2.f, 3.f, (float)bid};
    constexpr int LPR = 64 / ROWS, WPX = LPR * 4;
    const int r = lane / LPR, q = lane % LPR;
    const int rpw = OCB / 4;
    float *ob = out + ((size_t)n * OC + (size_t)ot * OCB + wave * rpw) * P + p0;
    for (int c0 = 0; c0 < TP; c0 += WPX)
        for (int r0 = 0; r0 < rpw; r0 += ROWS) {
            const int px = c0 + 4 * q;
            if (px < TP && p0 + px < P && r0 + r < rpw) *reinterpret_cast<vf4 *>(ob + (size_t)(r0 + r) * P + px) = v;
        }
}

template <int ROWS>
static float run(float *out, int N, int OC, int P, int TP, int OCB, int map)
{
    const int tiles_p = (P + TP - 1) / TP, n_oc = OC / OCB;
    const long units = (long)N * tiles_p;
    long blocks = (units + 7) / 8 * 8 * n_oc;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<ROWS>), dim3(blocks), dim3(256), 0, 0, out, N, OC, P, TP, OCB, tiles_p, n_oc, map, units);
    (void)hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k<ROWS>), dim3(blocks), dim3(256), 0, 0, out, N, OC, P, TP, OCB, tiles_p, n_oc, map, units);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main()
{
    struct S { int N, OC, P; } shapes[] = {{256, 256, 3136}, {256, 512, 784}, {256, 1024, 196}};
    const int tps[] = {128, 160, 224, 256, 512, 1024, 3136};
    const int ocbs[] = {128, 32, 16};
    for (auto s : shapes) {
        const size_t bytes = (size_t)s.N * s.OC * s.P * 4;
        float *out; (void)hipMalloc(&out, bytes + (64 << 20));
        printf("N=%d OC=%d P=%d (%.0f MB)\n", s.N, s.OC, s.P, bytes / 1e6);
        for (int ocb : ocbs)
            for (int tp : tps) {
                if (tp > s.P && tp != 224 && tp != 256) continue;
                if ((s.P % 4) != 0) continue;
                printf("  OCB=%3d TP=%4d:", ocb, tp);
                for (int map = 0; map < 4; ++map) {
                    const float t8 = run<8>(out, s.N, s.OC, s.P, tp, ocb, map);
                    const float t2 = run<2>(out, s.N, s.OC, s.P, tp, ocb, map);
                    printf("  m%d %.2f/%.2f", map, bytes / t8 / 1e9, bytes / t2 / 1e9);
                }
                printf("   TB/s (8x128B / 2x512B)\n");
            }
        (void)hipFree(out);
    }
    return 0;
}
