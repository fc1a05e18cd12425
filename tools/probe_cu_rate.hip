// probe_cu_rate.hip -- what ONE CU can move, by number of active CUs: stores, HBM loads, L2-resident loads.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe_cu_rate tools/probe_cu_rate.hip && tools/probe_cu_rate
// Each workgroup (256 or 512 threads, one per CU while grid <= 256) works on a private contiguous region with 16-byte
// accesses; the grid is swept so that per-CU limits and chip-wide (HBM) limits separate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_store(float4 *out, size_t per_wg_vec, int iters)
{
    float4 *base = out + (size_t)blockIdx.x * per_wg_vec;
    const float4 v = make_float4(1.f, 2.f, 3.f, (float)threadIdx.x);
    for (int it = 0; it < iters; ++it)
        for (size_t i = threadIdx.x; i < per_wg_vec; i += blockDim.x) base[i] = v;
}

__global__ void k_load(const float4 *in, size_t per_wg_vec, int iters, float *sink, int shared_region)
{
    const float4 *base = in + (shared_region ? 0 : (size_t)blockIdx.x * per_wg_vec);
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        for (size_t i = threadIdx.x; i < per_wg_vec; i += 4 * blockDim.x) {
            float4 a = base[i], b = base[(i + blockDim.x) % per_wg_vec], c = base[(i + 2 * blockDim.x) % per_wg_vec], d = base[(i + 3 * blockDim.x) % per_wg_vec];
            acc += a.x + b.y + c.z + d.w;
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

int main()
{
    const size_t region = 4u << 20;                 // 4 MiB per workgroup
    const int max_wg = 512;
    float4 *buf;
    float *sink;
    CK(hipMalloc(&buf, region * max_wg));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 0, region * max_wg));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t vec = region / 16;
    for (int threads : {256, 512}) {
        for (int wg : {8, 32, 64, 128, 256, 512}) {
            float ms;
            // stores
            hipLaunchKernelGGL(k_store, dim3(wg), dim3(threads), 0, 0, buf, vec, 1);
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_store, dim3(wg), dim3(threads), 0, 0, buf, vec, 4);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            const double st = 4.0 * region * wg / (ms * 1e-3) / 1e9;
            // HBM loads (private 4 MiB regions, 4 passes: the first misses everywhere when wg * 4 MiB > caches)
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_load, dim3(wg), dim3(threads), 0, 0, buf, vec, 4, sink, 0);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            const double ld = 4.0 * region * wg / (ms * 1e-3) / 1e9;
            // L2-resident loads: every workgroup reads the same 1 MiB, 16 passes
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_load, dim3(wg), dim3(threads), 0, 0, buf, (size_t)(1u << 20) / 16, 16, sink, 1);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            const double l2 = 16.0 * (1u << 20) * wg / (ms * 1e-3) / 1e9;
            const int cus = wg < 256 ? wg : 256;
            printf("threads %3d  wg %3d : store %7.1f GB/s (%5.1f per CU)   load %7.1f GB/s (%5.1f per CU)   L2 load %8.1f GB/s (%6.1f per CU)\n",
                   threads, wg, st, st / cus, ld, ld / cus, l2, l2 / cus);
        }
    }
    return 0;
}
