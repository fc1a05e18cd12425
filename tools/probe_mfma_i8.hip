// probe_mfma_i8.hip -- checks, with exact integer data on the GPU, the operand/result lane maps
// this project assumes for v_mfma_i32_32x32x32_i8 (guide: "Other dtypes: check the map with
// exact integer data before relying on it").  Build: hipcc --offload-arch=gfx950 -O2 -o probe ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void probe(const int8_t *A /*[32][32] row-major: A[m][k]*/, const int8_t *B /*[32][32]: B[k][n]*/,
                      int *D /*[32][32]*/)
{
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    int8_t a[16], b[16];
    for (int j = 0; j < 16; ++j) {
        a[j] = A[r * 32 + 16 * h + j];      // lane (row r, half h) holds A[r][16h + j]
        b[j] = B[(16 * h + j) * 32 + r];    // lane (col r, half h) holds B[16h + j][r]
    }
    v4i av, bv;
    __builtin_memcpy(&av, a, 16);
    __builtin_memcpy(&bv, b, 16);
    v16i c = {0};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, c, 0, 0, 0);
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * h;  // guide: C/D map of the 32x32 shapes
        D[row * 32 + r] = c[reg];
    }
}

int main()
{
    int8_t hA[1024], hB[1024];
    int hD[1024], ref[1024];
    srand(1);
    for (int i = 0; i < 1024; ++i) { hA[i] = (int8_t)(rand() % 256 - 128); hB[i] = (int8_t)(rand() % 256 - 128); }
    for (int m = 0; m < 32; ++m)
        for (int n = 0; n < 32; ++n) {
            int s = 0;
            for (int k = 0; k < 32; ++k) s += (int)hA[m * 32 + k] * (int)hB[k * 32 + n];
            ref[m * 32 + n] = s;
        }
    int8_t *dA, *dB; int *dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; ++i) bad += hD[i] != ref[i];
    printf("mfma_i32_32x32x32_i8 map check: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
    return bad != 0;
}
