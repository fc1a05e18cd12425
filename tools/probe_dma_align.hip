// probe_dma_align.hip -- does global_load_lds_dwordx4 (LDS-DMA, 16 B per lane) accept source addresses that are only
// 4-byte or 1-byte aligned on gfx950, and at what cost?  Needed to fill a padded [channel][pixel] LDS image straight
// from NCHW planes of 196 bytes (14x14: rows start 4-byte aligned) and 49 bytes (7x7: byte aligned) without a
// register round trip.  Each lane copies 16 bytes from src + off + i * stride into LDS slot i; the block then dumps
// its LDS image so the host can compare byte for byte.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ __launch_bounds__(256) void k_dma(const uint8_t *src, uint4 *dst, int off, int stride, int reps)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[256 * 16];
    const int tid = threadIdx.x, wave = tid >> 6;
    const size_t i = (size_t)blockIdx.x * 256 + tid;
    for (int r = 0; r < reps; ++r)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + off + i * stride),
                                         (__attribute__((address_space(3))) void *)(lds + wave * 1024), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
    __syncthreads();
    dst[i] = *reinterpret_cast<const uint4 *>(lds + tid * 16);
}

int main()
{
    const int n = 1 << 20;
    std::vector<uint8_t> h((size_t)n * 64 + 64);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint8_t)(i * 131 + (i >> 8));
    uint8_t *src; uint4 *dst;
    (void)hipMalloc(&src, h.size()); (void)hipMalloc(&dst, (size_t)n * 16);
    (void)hipMemcpy(src, h.data(), h.size(), hipMemcpyHostToDevice);
    std::vector<uint4> out(n);
    for (int stride : {16, 49, 196}) for (int off : {0, 4, 8, 12, 1, 2, 3, 7, 13}) {
        if ((size_t)off + (size_t)n * stride + 16 > h.size()) continue;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(k_dma, dim3(n / 256), dim3(256), 0, 0, src, dst, off, stride, 1);
        (void)hipEventRecord(e0);
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k_dma, dim3(n / 256), dim3(256), 0, 0, src, dst, off, stride, 1);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(out.data(), dst, (size_t)n * 16, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < n; ++i) if (memcmp(&out[i], &h[off + (size_t)i * stride], 16)) ++bad;
        printf("lds-dma stride %3d off %2d: %s (%d bad)  %.4f ms\n", stride, off, bad ? "WRONG" : "ok", bad, ms / 10);
    }
    return 0;
}
