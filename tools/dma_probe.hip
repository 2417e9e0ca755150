// Probe: how fast can LDS-DMA (global_load_lds_dwordx4) fill LDS per CU, by access shape / sharing / grid size?
// Standalone diagnostic (hipcc --offload-arch=gfx950 tools/dma_probe.hip -o tools/diag/dma_probe).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
#define DMA16(src, dst) __builtin_amdgcn_global_load_lds((glb_void_t*)(src), (lds_void_t*)(dst), 16, 0, 0)

// each block streams `iters` stages of 32 KiB (8 waves x 4 pieces of 1 KiB) with 3 stages in flight.
// mode 0: piece = 16 rows x 64 B at row stride `ld` bytes; mode 1: piece = 8 rows x 128 B; mode 2: 1 KiB contiguous
template <int MODE>
__global__ __launch_bounds__(512) void probe(const char* base, size_t block_stride, size_t ld, int iters, int share,
                                             unsigned long long* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int blk = share ? (blockIdx.x % share) : blockIdx.x;
    const char* src = base + (size_t)blk * block_stride;
    size_t off[4];
    for (int j = 0; j < 4; ++j) {
        int piece = wid * 4 + j;   // 32 pieces per stage
        if (MODE == 0) off[j] = (size_t)(piece * 16 + (lane >> 2)) * ld + (lane & 3) * 16;
        else if (MODE == 1) off[j] = (size_t)(piece * 8 + (lane >> 3)) * ld + (lane & 7) * 16;
        else off[j] = (size_t)piece * 1024 + lane * 16;
    }
    const size_t kstep = MODE == 0 ? 64 : (MODE == 1 ? 128 : 32768);
    unsigned long long t0 = 0, t1 = 0;
    if (threadIdx.x == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int s = 0; s < 3; ++s)
        for (int j = 0; j < 4; ++j) DMA16(src + off[j] + s * kstep, smem + s * 32768 + (wid * 4 + j) * 1024);
    for (int t = 0; t < iters; ++t) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int j = 0; j < 4; ++j) DMA16(src + off[j] + (size_t)(t + 3) * kstep, smem + ((t + 3) & 3) * 32768 + (wid * 4 + j) * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (threadIdx.x == 0) {
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        out[blockIdx.x] = t1 - t0;
    }
}

int main() {
    const size_t bytes = (size_t)3 << 30;
    char* buf; hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes);
    unsigned long long* out; hipMalloc(&out, 4096 * 8);
    const int iters = 21;
    struct Cfg { int mode; size_t ld; const char* name; } cfgs[] = {
        {0, 1536, "16x64B rows, ld 1536"}, {1, 1536, "8x128B rows, ld 1536"}, {0, 6144, "16x64B rows, ld 6144"}, {2, 0, "1KiB contiguous"}};
    for (auto& c : cfgs)
        for (int share : {0, 1, 12})
            for (int grid : {64, 256, 512, 2048}) {
                // per block footprint: rows * ld (mode 0/1: 512 or 256 rows) ; keep inside the buffer
                size_t rows = c.mode == 0 ? 512 : 256;
                size_t bstride = c.mode == 2 ? (size_t)(iters + 4) * 32768 : rows * c.ld;
                if (bstride * (share ? share : grid) + (iters + 4) * 128 > bytes) continue;
                for (int rep = 0; rep < 2; ++rep) {
                    if (c.mode == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(512), 131072, 0, buf, bstride, c.ld, iters, share, out);
                    if (c.mode == 1) hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(512), 131072, 0, buf, bstride, c.ld, iters, share, out);
                    if (c.mode == 2) hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(512), 131072, 0, buf, bstride, c.ld, iters, share, out);
                    hipDeviceSynchronize();
                }
                std::vector<unsigned long long> h(grid);
                hipMemcpy(h.data(), out, grid * 8, hipMemcpyDeviceToHost);
                double sum = 0; for (auto v : h) sum += v;
                double cyc = sum / grid, kb = (iters + 3) * 32.0;
                printf("%-24s share=%2d grid=%4d: %8.0f ticks/block  %6.1f B/tick/CU\n", c.name, share, grid, cyc, kb * 1024 / cyc);
            }
    return 0;
}
