// Probe: how fast does ONE CU retire output stores?  Each workgroup (512 threads, 160 KiB of dynamic LDS so that it owns its CU) writes
// `bytes` of its own region with 16-byte stores in whole 256-B row segments (the GEMM epilogues' pattern), non-temporal or regular, and
// times issue (s_memtime before the first / after the last store) and completion (after s_waitcnt vmcnt(0)) in shader cycles.
//   hipcc --offload-arch=gfx950 -O3 tools/store_probe.hip -o tools/diag/store_probe && tools/diag/store_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(512) void probe(float* out, size_t bytes_per_wg, unsigned long long* t) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x;
    char* base = (char*)out + (size_t)blockIdx.x * bytes_per_wg;
    const int n = (int)(bytes_per_wg / (512 * 16));
    unsigned long long t0 = 0, t1 = 0, t2 = 0;
    const f32x4 v = {1.f, 2.f, 3.f, (float)tid};
    __builtin_amdgcn_s_barrier();
    if (tid == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int i = 0; i < n; ++i) {
        f32x4* dst = (f32x4*)(base + ((size_t)i * 512 + tid) * 16);
        if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
    }
    if (tid == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (tid == 0) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory"); t[2 * blockIdx.x] = t1 - t0; t[2 * blockIdx.x + 1] = t2 - t0; }
    if (smem[tid] == 77) out[0] = 0.f;   // keep the LDS allocation
}
template <bool NT>
void run(const char* name, int wgs, size_t bytes, float* buf, unsigned long long* t) {
    (void)hipFuncSetAttribute((const void*)probe<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    std::vector<unsigned long long> h(2 * wgs);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(probe<NT>, dim3(wgs), dim3(512), 160 * 1024, 0, buf, bytes, t);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h.data(), t, sizeof(unsigned long long) * 2 * wgs, hipMemcpyDeviceToHost);
    std::vector<double> iss, don;
    for (int i = 0; i < wgs; ++i) { iss.push_back((double)h[2 * i]); don.push_back((double)h[2 * i + 1]); }
    std::sort(iss.begin(), iss.end()); std::sort(don.begin(), don.end());
    printf("%-12s %3d workgroups x %4zu KB: issued in %7.0f ticks (%5.1f B/tick), complete after %7.0f ticks (%5.1f B/tick per CU); launch %.3f ms = %.2f TB/s\n",
           name, wgs, bytes >> 10, iss[wgs / 2], bytes / iss[wgs / 2], don[wgs / 2], bytes / don[wgs / 2], ms, wgs * (double)bytes / ms / 1e9);
}
int main() {
    float* buf; unsigned long long* t;
    const size_t bytes = 512 << 10;
    (void)hipMalloc(&buf, 256 * (4 << 20)); (void)hipMalloc(&t, 8192);
    for (int wgs : {1, 8, 32, 64, 128, 256}) { run<true>("non-temporal", wgs, bytes, buf, t); run<false>("regular", wgs, bytes, buf, t); }
    for (size_t b : {(size_t)128 << 10, (size_t)2 << 20}) { run<true>("non-temporal", 256, b, buf, t); run<true>("non-temporal", 48, b, buf, t); }
    return 0;
}
