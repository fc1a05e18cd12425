// probe_unaligned.hip -- do global_load_dwordx4 / global_store_dwordx4 work at byte-unaligned (and dword-aligned but
// not 16-byte aligned) addresses on gfx950, and what do they cost?  (needed for 49-byte planes: 7x7 feature maps)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void k_load(const uint8_t *src, uint4 *dst, int off, int stride)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint4 v;
    __builtin_memcpy(&v, src + off + (size_t)i * stride, 16);
    dst[i] = v;
}
__global__ void k_store(uint8_t *dst, int off, int stride)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float4 v = make_float4(i, i + 0.25f, i + 0.5f, i + 0.75f);
    __builtin_memcpy(dst + off + (size_t)i * stride, &v, 16);
}
int main()
{
    const int n = 1 << 20;
    std::vector<uint8_t> h((size_t)n * 64 + 64);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint8_t)(i * 131 + (i >> 8));
    uint8_t *src; uint4 *dst; uint8_t *st;
    (void)hipMalloc(&src, h.size()); (void)hipMalloc(&dst, (size_t)n * 16); (void)hipMalloc(&st, h.size());
    (void)hipMemcpy(src, h.data(), h.size(), hipMemcpyHostToDevice);
    std::vector<uint4> out(n);
    for (int stride : {16, 49}) for (int off : {0, 1, 2, 3, 4, 7, 8, 13}) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k_load, dim3(n / 256), dim3(256), 0, 0, src, dst, off, stride);
        (void)hipEventRecord(e0);
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k_load, dim3(n / 256), dim3(256), 0, 0, src, dst, off, stride);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(out.data(), dst, (size_t)n * 16, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < n; ++i) if (memcmp(&out[i], &h[off + (size_t)i * stride], 16)) ++bad;
        printf("load  stride %2d off %2d: %s  %.4f ms\n", stride, off, bad ? "WRONG" : "ok", ms / 10);
    }
    for (int stride : {16, 196}) for (int off : {0, 4, 8, 12, 2}) {
        (void)hipMemset(st, 0, h.size());
        const int m = stride == 16 ? n : n / 8;
        hipLaunchKernelGGL(k_store, dim3(m / 256), dim3(256), 0, 0, st, off, stride);
        std::vector<uint8_t> back(h.size());
        (void)hipMemcpy(back.data(), st, h.size(), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < m; ++i) { float v[4]; memcpy(v, &back[off + (size_t)i * stride], 16); if (v[0] != i || v[3] != i + 0.75f) ++bad; }
        printf("store stride %3d off %2d: %s\n", stride, off, bad ? "WRONG" : "ok");
    }
    return 0;
}
