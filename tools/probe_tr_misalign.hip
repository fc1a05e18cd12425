// probe_tr_misalign.hip -- does ds_read_b64_tr_b8 take addresses that are not 8-byte aligned, and what does it cost?
// (a 3x3 tap shifts the pixel run of a [channel][row][col] LDS image by kx = 0, 1, 2 BYTES.)
// Rows of PITCH bytes; lane i of a 16-lane group reads row i/2, half i%2, shifted by `mis` bytes; group g reads pixels
// 16 g .. 16 g + 15.  Expected: result byte r of lane i = lds[row r][16 g + i + mis].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v2i __attribute__((ext_vector_type(2)));
#define PITCH 96
__global__ void probe(uint32_t *out, long long *cyc, int mis, int iters)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[8192];
    const int l = threadIdx.x;
    for (int i = l; i < 8192; i += 64) lds[i] = (uint8_t)((i * 7 + (i >> 8)) & 255);
    __syncthreads();
    const int g = l >> 4, i = l & 15;
    const uint32_t addr = (uint32_t)(uintptr_t)lds + (i >> 1) * PITCH + 8 * (i & 1) + 16 * g + mis;
    v2i r0, r1, r2, r3, a0 = {0, 0}, a1 = {0, 0};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        asm volatile("ds_read_b64_tr_b8 %0, %4 offset:0\n\tds_read_b64_tr_b8 %1, %4 offset:768\n\t"
                     "ds_read_b64_tr_b8 %2, %4 offset:1536\n\tds_read_b64_tr_b8 %3, %4 offset:2304\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(addr) : "memory");
        a0 += r0 ^ r2; a1 += r1 ^ r3;
    }
    const long long t1 = clock64();
    asm volatile("ds_read_b64_tr_b8 %0, %1 offset:0\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r0) : "v"(addr) : "memory");
    out[l * 2] = (uint32_t)r0[0];
    out[l * 2 + 1] = (uint32_t)r0[1];
    if (l == 0) cyc[0] = t1 - t0;
    if (a0[0] == 0x12345678 && a1[1] == 0x7654321) out[0] = 0;
}
// the same with plain ds_read_b64 (reference cost of an aligned / misaligned 8-byte read)
__global__ void probe_plain(long long *cyc, int mis, int iters)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[8192];
    const int l = threadIdx.x;
    for (int i = l; i < 8192; i += 64) lds[i] = (uint8_t)i;
    __syncthreads();
    const int g = l >> 4, i = l & 15;
    const uint32_t addr = (uint32_t)(uintptr_t)lds + (i >> 1) * PITCH + 8 * (i & 1) + 16 * g + mis;
    v2i r0, r1, r2, r3, a0 = {0, 0};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        asm volatile("ds_read_b64 %0, %4 offset:0\n\tds_read_b64 %1, %4 offset:768\n\t"
                     "ds_read_b64 %2, %4 offset:1536\n\tds_read_b64 %3, %4 offset:2304\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(addr) : "memory");
        a0 += r0 ^ r2 ^ r1 ^ r3;
    }
    const long long t1 = clock64();
    if (l == 0) cyc[0] = t1 - t0;
    if (a0[0] == 0x12345678) cyc[1] = 0;
}
int main()
{
    uint32_t *d, h[128];
    long long *c, hc[2];
    (void)hipMalloc(&d, 512); (void)hipMalloc(&c, 16);
    for (int mis = 0; mis < 9; ++mis) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, c, mis, 2000);
        (void)hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int b = 0; b < 8; ++b) {
            const int got = (h[l * 2 + b / 4] >> (8 * (b % 4))) & 255;
            const int a = b * PITCH + 16 * (l >> 4) + (l & 15) + mis;
            const int want = (a * 7 + (a >> 8)) & 255;
            bad += got != want;
        }
        hipLaunchKernelGGL(probe_plain, dim3(1), dim3(64), 0, 0, c, mis, 2000);
        long long hp[2];
        (void)hipMemcpy(hp, c, 16, hipMemcpyDeviceToHost);
        printf("mis %d: tr_b8 wrong bytes %d / 512, %.1f cycles per 4 reads; plain b64 %.1f cycles per 4 reads\n", mis, bad,
               hc[0] / 2000.0, hp[0] / 2000.0);
        if (bad) {
            for (int l = 0; l < 4; ++l) {
                printf("  lane %d got:", l);
                for (int b = 0; b < 8; ++b) printf(" %3d", (h[l * 2 + b / 4] >> (8 * (b % 4))) & 255);
                printf("\n");
            }
        }
    }
    return 0;
}
