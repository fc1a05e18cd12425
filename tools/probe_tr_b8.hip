// probe_tr_b8.hip -- empirical semantics of ds_read_b64_tr_b8 on gfx950 (the ISA document is not in
// this image).  LDS byte a holds an id of its own address; every lane supplies an 8-byte aligned address
// (two different lane->address maps) and we print, per lane, which LDS addresses its 8 result bytes came from.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v2i __attribute__((ext_vector_type(2)));

__global__ void probe(uint32_t *out, int map, int hi)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[4096];
    const int l = threadIdx.x;
    for (int i = l; i < 4096; i += 64) lds[i] = (uint8_t)(hi ? (i >> 8) : (i & 255));
    __syncthreads();
    const int g = l >> 4, i = l & 15;
    int addr;
    if (map == 0) addr = (i >> 1) * 64 + 8 * (i & 1) + 16 * g;   // rows of 64 B: lane i -> row i/2, half i%2; group g -> +16 cols
    else if (map == 1) addr = i * 64 + 8 * g;                     // lane i -> row i, group g -> +8 cols
    else addr = l * 8;                                            // flat: lane l -> bytes 8l..8l+7
    v2i r = __builtin_amdgcn_ds_read_tr8_b64_v2i32((v2i __attribute__((address_space(3))) *)(lds + addr));
    out[l * 2] = (uint32_t)r[0];
    out[l * 2 + 1] = (uint32_t)r[1];
}

int main()
{
    uint32_t *d, h0[128], h1[128];
    hipMalloc(&d, 512);
    for (int map = 0; map < 3; ++map) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, map, 0);
        hipMemcpy(h0, d, 512, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, map, 1);
        hipMemcpy(h1, d, 512, hipMemcpyDeviceToHost);
        printf("map %d\n", map);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int b = 0; b < 8; ++b) {
                const int lo = (h0[l * 2 + b / 4] >> (8 * (b % 4))) & 255, hi = (h1[l * 2 + b / 4] >> (8 * (b % 4))) & 255;
                printf(" %4d", hi * 256 + lo);
            }
            printf("\n");
        }
    }
    return 0;
}
