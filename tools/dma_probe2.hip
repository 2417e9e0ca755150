// Probe 2: LDS-DMA fill rate in the GEMM access pattern (block (tm,tn) streams A panel tm and B panel tn along K),
// no MFMA, no epilogue.  Piece shapes: MODE 0 = 16 rows x 64 B (BK 32 stages), MODE 1 = 8 rows x 128 B (BK 64 stages).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
#define DMA16(src, dst) __builtin_amdgcn_global_load_lds((glb_void_t*)(src), (lds_void_t*)(dst), 16, 0, 0)
__device__ __forceinline__ int xcd_remap(int bid, int nb) {
    int q = nb >> 3, r = nb & 7, x = bid & 7, slot = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + slot;
}
template <int MODE>
__global__ __launch_bounds__(512) void probe(const char* A, const char* B, int tiles_n, int K, int remap, int ngroup,
                                             int tiles_m, unsigned long long* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int logical = remap ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    int tm, tn;
    if (ngroup) {
        int per = tiles_m * ngroup, g = logical / per, rem = logical - g * per;
        int gw = tiles_n - g * ngroup < ngroup ? tiles_n - g * ngroup : ngroup;
        tm = rem / gw; tn = g * ngroup + rem % gw;
    } else { tm = logical / tiles_n; tn = logical % tiles_n; }
    const size_t ld = (size_t)K * 2;
    const char* a = A + (size_t)tm * 256 * ld;
    const char* b = B + (size_t)tn * 256 * ld;
    // a stage = 32 KiB: MODE 0: A 256 rows x 64 B + B 256 rows x 64 B ; MODE 1: (half stage) A 128.. use 64 KiB stages of 2 slots
    size_t offa[4], offb[4];
    int npa;
    if (MODE == 0) { npa = 2; for (int j = 0; j < 2; ++j) { offa[j] = (size_t)(wid * 32 + 16 * j + (lane >> 2)) * ld + (lane & 3) * 16; offb[j] = offa[j]; } }
    else { npa = 4; for (int j = 0; j < 4; ++j) { offa[j] = (size_t)(wid * 32 + 8 * j + (lane >> 3)) * ld + (lane & 7) * 16; offb[j] = offa[j]; } }
    const int kstep = MODE == 0 ? 64 : 128;
    const int nt = (int)(ld / kstep);
    const int NS = MODE == 0 ? 4 : 2;                 // ring slots (32 KiB / 64 KiB)
    const int slotb = MODE == 0 ? 32768 : 65536;
    unsigned long long t0 = 0, t1 = 0;
    if (threadIdx.x == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    auto issue = [&](int t) {
        char* s = smem + (t % NS) * slotb + wid * npa * 1024;
        for (int j = 0; j < npa; ++j) { DMA16(a + offa[j] + (size_t)t * kstep, s + j * 1024); DMA16(b + offb[j] + (size_t)t * kstep, s + slotb / 2 + j * 1024); }
    };
    for (int s = 0; s < NS - 1; ++s) issue(s);
    for (int t = 0; t < nt; ++t) {
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + NS - 1 < nt) issue(t + NS - 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (threadIdx.x == 0) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory"); out[blockIdx.x] = t1 - t0; }
}
int main() {
    const int M = 78848;
    char *A, *B; unsigned long long* out;
    hipMalloc(&A, (size_t)M * 3072 * 2 + 65536); hipMalloc(&B, (size_t)3072 * 3072 * 2 + 65536); hipMalloc(&out, 8192 * 8);
    hipMemset(A, 1, (size_t)M * 3072 * 2); hipMemset(B, 1, (size_t)3072 * 3072 * 2);
    struct S { const char* n; int N, K; } shapes[] = {{"qkv", 2304, 768}, {"out", 768, 768}, {"fc", 3072, 768}, {"proj", 768, 3072}};
    for (auto& sh : shapes)
        for (int mode = 0; mode < 2; ++mode)
            for (int remap = 0; remap < 2; ++remap)
                for (int ng : {0, 4}) {
                    int tiles_m = M / 256, tiles_n = sh.N / 256, grid = tiles_m * tiles_n;
                    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                    float ms = 0;
                    for (int rep = 0; rep < 3; ++rep) {
                        hipEventRecord(e0);
                        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(512), 131072, 0, A, B, tiles_n, sh.K, remap, ng, tiles_m, out);
                        else hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(512), 131072, 0, A, B, tiles_n, sh.K, remap, ng, tiles_m, out);
                        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
                    }
                    std::vector<unsigned long long> h(grid);
                    hipMemcpy(h.data(), out, grid * 8, hipMemcpyDeviceToHost);
                    double sum = 0; for (auto v : h) sum += v;
                    double bytes = 2.0 * 256 * sh.K * 2;
                    printf("%-5s piece=%s remap=%d ngroup=%d: kernel %.3f ms  %7.0f ticks/block  %5.1f B/tick/CU  (%.1f TB/s)\n", sh.n, mode ? "8x128B" : "16x64B", remap, ng,
                           ms, sum / grid, bytes / (sum / grid), bytes * grid / ms / 1e9);
                }
    return 0;
}
