// probe_dma_mask.hip -- two questions about global_load_lds_dwordx4 on gfx950:
//  (a) do EXEC-masked lanes leave their LDS slot untouched?
//  (b) does a ds_write issued by the same lane AFTER `s_waitcnt vmcnt(0)` win over the DMA's own write to that slot?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ __launch_bounds__(256) void k(const uint8_t *src, uint4 *dst, int mode)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[256 * 16];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    *reinterpret_cast<uint4 *>(lds + tid * 16) = make_uint4(0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu, 0xAAAAAAAAu);
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * 256 + tid;
    if (mode == 0) {
        if (lane % 3 != 1)     // lanes 1, 4, 7, ... masked off
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + i * 16),
                                             (__attribute__((address_space(3))) void *)(lds + wave * 1024), 16, 0, 0);
        __builtin_amdgcn_s_waitcnt(0x0f70);
    } else {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + i * 16),
                                         (__attribute__((address_space(3))) void *)(lds + wave * 1024), 16, 0, 0);
        __builtin_amdgcn_s_waitcnt(0x0f70);
        if (lane % 3 == 1) *reinterpret_cast<uint32_t *>(lds + tid * 16) = 0x12345678u;
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
    __syncthreads();
    dst[i] = *reinterpret_cast<const uint4 *>(lds + tid * 16);
}

int main()
{
    const int n = 256 * 1024;
    std::vector<uint8_t> h((size_t)n * 16);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint8_t)(i * 131 + (i >> 8));
    uint8_t *src; uint4 *dst;
    (void)hipMalloc(&src, h.size()); (void)hipMalloc(&dst, (size_t)n * 16);
    (void)hipMemcpy(src, h.data(), h.size(), hipMemcpyHostToDevice);
    std::vector<uint4> out(n);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, src, dst, mode);
        (void)hipMemcpy(out.data(), dst, (size_t)n * 16, hipMemcpyDeviceToHost);
        long dma = 0, kept = 0, patched = 0, other = 0;
        for (int i = 0; i < n; ++i) {
            if ((i & 63) % 3 != 1) { if (memcmp(&out[i], &h[(size_t)i * 16], 16)) ++other; continue; }
            uint32_t w0 = out[i].x;
            if (!memcmp(&out[i], &h[(size_t)i * 16], 16)) ++dma;
            else if (w0 == 0xAAAAAAAAu) ++kept;
            else if (w0 == 0x12345678u && !memcmp(&out[i].y, &h[(size_t)i * 16 + 4], 12)) ++patched;
            else ++other;
        }
        printf("mode %d (%s): special lanes -> dma data %ld, untouched %ld, patched %ld, other/wrong %ld\n", mode,
               mode == 0 ? "EXEC-masked DMA lanes" : "ds_write after vmcnt(0)", dma, kept, patched, other);
    }
    return 0;
}
