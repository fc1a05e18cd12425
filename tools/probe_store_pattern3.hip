// probe_store_pattern3.hip -- write rate vs contiguous run length.  The tensor is [ROWS_TOTAL][RL bytes]
// (think (n, oc) rows of P*4 bytes); a 256-thread workgroup writes 64 KB as (64 KB / B) row pieces of B bytes
// (R consecutive rows x one B-byte column slot).  B = RL reproduces a fully linear 64 KB block.
// map 0: neighbouring column slots on different XCDs;  map 3: each XCD owns a contiguous range of blocks.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float vf4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k(char *out, long rows_total, int RL, int B, int map, long nblocks)
{
    const int tid = threadIdx.x;
    long bid = blockIdx.x;
    if (map == 3) { const long per = (nblocks + 7) / 8; bid = (bid & 7) * per + (bid >> 3); if (bid >= nblocks) return; }
    const int R = 65536 / B;                  // rows per block
    const int slots = RL / B;                 // column slots per row
    const long rb = bid / slots; const int slot = (int)(bid - rb * slots);   // slot fastest
    char *base = out + (rb * R) * (long)RL + (long)slot * B;
    const vf4 v = {1.f, 2.f, 3.f, (float)bid};
    // 256 threads x 16 B = 4 KB per step; 16 steps.  thread t of step s covers byte (s*4096 + t*16) of the
    // block's (row-major) R x B region
    for (int s = 0; s < 16; ++s) {
        const int off = s * 4096 + tid * 16;
        const int r = off / B, c = off - r * B;
        *reinterpret_cast<vf4 *>(base + (long)r * RL + c) = v;
    }
}

int main()
{
    const int RLs[] = {16384, 12544, 3136, 784};
    const size_t total = 768u << 20;
    char *out; (void)hipMalloc(&out, total + (64 << 20));
    for (int RL : RLs) {
        const int Bs[] = {64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, RL};
        printf("row length %d B\n", RL);
        for (int B : Bs) {
            if (B > RL || RL % B || 65536 % B) continue;
            const int R = 65536 / B;
            const long rows_total = (long)(total / RL) / R * R;
            const long nblocks = rows_total / R * (RL / B);
            printf("  B=%5d (R=%4d): ", B, R);
            for (int map : {0, 3}) {
                hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
                const long grid = (nblocks + 7) / 8 * 8;
                for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, rows_total, RL, B, map, nblocks);
                (void)hipEventRecord(e0);
                for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, rows_total, RL, B, map, nblocks);
                (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 20;
                printf("  m%d %.2f TB/s", map, (double)nblocks * 65536 / ms / 1e9);
            }
            printf("\n");
        }
    }
    return 0;
}
