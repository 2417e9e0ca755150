// Probe: does data one kernel has just WRITTEN come back faster (Infinity Cache / MALL hits) when the next kernel reads it?  Kernel W writes
// N MB (16-byte stores, non-temporal or regular), kernel R reads the same N MB (forward, or last-written-first) and sums it.  Prints the
// read kernel's bandwidth; a cold read of a different buffer of the same size is the reference.
//   hipcc --offload-arch=gfx950 -O3 tools/mall_probe.hip -o tools/diag/mall_probe && tools/diag/mall_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void wr(f32x4* p, size_t n16) {
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v;
    }
}
__global__ __launch_bounds__(256) void rd(const f32x4* p, size_t n16, int reverse, float* out) {
    float s = 0.f;
    // blocks walk contiguous 64-KB chunks; reverse: the chunk order is mirrored (last-written data first)
    const size_t chunk = 4096, nchunk = n16 / chunk;
    for (size_t c = blockIdx.x; c < nchunk; c += gridDim.x) {
        const size_t cc = reverse ? nchunk - 1 - c : c;
        for (size_t i = threadIdx.x; i < chunk; i += 256) { const f32x4 v = p[cc * chunk + i]; s += v.x + v.w; }
    }
    if (s == 12345.f) out[0] = s;
}
int main() {
    f32x4 *a, *b; float* out;
    const size_t maxb = (size_t)1 << 30;
    (void)hipMalloc(&a, maxb); (void)hipMalloc(&b, maxb); (void)hipMalloc(&out, 64);
    (void)hipMemset(b, 0, maxb);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (size_t mb : {32, 64, 128, 192, 256, 512, 1024}) {
        const size_t n16 = (mb << 20) / 16;
        for (int nt = 0; nt < 2; ++nt)
            for (int rev = 0; rev < 2; ++rev) {
                float best = 1e9f, cold = 1e9f;
                for (int rep = 0; rep < 3; ++rep) {
                    // cold reference: read the OTHER buffer (evicts), then write a, then read a
                    (void)hipEventRecord(e0);
                    hipLaunchKernelGGL(rd, dim3(2048), dim3(256), 0, 0, b, n16, rev, out);
                    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                    float ms; (void)hipEventElapsedTime(&ms, e0, e1); cold = ms < cold ? ms : cold;
                    if (nt) hipLaunchKernelGGL(wr<true>, dim3(2048), dim3(256), 0, 0, a, n16); else hipLaunchKernelGGL(wr<false>, dim3(2048), dim3(256), 0, 0, a, n16);
                    (void)hipEventRecord(e0);
                    hipLaunchKernelGGL(rd, dim3(2048), dim3(256), 0, 0, a, n16, rev, out);
                    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                    (void)hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
                }
                printf("%5zu MB  written %-12s read %-8s: %6.2f TB/s   (same size, not just written: %6.2f TB/s)\n", mb, nt ? "non-temporal" : "regular",
                       rev ? "reversed" : "forward", (mb << 20) / best / 1e9, (mb << 20) / cold / 1e9);
            }
    }
    return 0;
}
