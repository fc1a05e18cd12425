// probe_mfma_rate.hip -- issue rate of the int8 MFMA shapes on gfx950, one wave per SIMD, operands in
// registers, 7 independent accumulators (the shape of the conv kernels' inner loop without LDS).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void rate(unsigned long long *out, int iters, int *sink)
{
    v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, (int)blockIdx.x, 8};
    v16i acc[7]; v4i acc4[7]; v16f accf[7];
    for (int t = 0; t < 7; ++t) { for (int r = 0; r < 16; ++r) { acc[t][r] = 0; accf[t][r] = 0; } for (int r = 0; r < 4; ++r) acc4[t][r] = 0; }
    v8s ah = {1,2,3,4,5,6,7,8}, bh = {8,7,6,5,4,3,2,1};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            if (MODE == 0) acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[t], 0, 0, 0);
            if (MODE == 1) acc4[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc4[t], 0, 0, 0);
            if (MODE == 2) accf[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, accf[t], 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int t = 0; t < 7; ++t) s += acc[t][0] + acc4[t][0] + (int)accf[t][0];
    if (s == 0x12345678) *sink = s;
    if (threadIdx.x % 64 == 0) out[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}

int main()
{
    unsigned long long *d, h[1024]; int *sink;
    hipMalloc(&d, sizeof(h)); hipMalloc(&sink, 4);
    const int iters = 2000;
    const char *names[3] = {"i32_32x32x32_i8", "i32_16x16x64_i8", "f32_32x32x16_bf16"};
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(256), dim3(256), 0, 0, d, iters, sink);
            if (mode == 1) hipLaunchKernelGGL(rate<1>, dim3(256), dim3(256), 0, 0, d, iters, sink);
            if (mode == 2) hipLaunchKernelGGL(rate<2>, dim3(256), dim3(256), 0, 0, d, iters, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            double cyc = 0; for (int i = 0; i < 1024; ++i) cyc += (double)h[i]; cyc /= 1024;
            const double n = 7.0 * iters;
            const double macs = (mode == 1 ? 16.0 * 16 * 64 : (mode == 0 ? 32.0 * 32 * 32 : 32.0 * 32 * 16));
            if (rep) printf("%-20s %.1f memtime-ticks per MFMA, kernel %.3f ms -> %.2f P(FL)OP/s chip-wide\n", names[mode], cyc / n, ms,
                            2.0 * macs * n * 1024 / (ms * 1e-3) / 1e15);
        }
    }
    return 0;
}
