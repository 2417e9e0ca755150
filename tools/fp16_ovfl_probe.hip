// Probe: does MODE.FP16_OVFL (hwreg MODE bit 23) make v_cvt_pk_f16_f32 / v_cvt_f16_f32 saturate at +-65504 instead of overflowing to inf on gfx950?
//   hipcc --offload-arch=gfx950 -O3 tools/fp16_ovfl_probe.hip -o tools/diag/fp16_ovfl_probe && tools/diag/fp16_ovfl_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__global__ void probe(const float* in, unsigned* out, int on) {
    if (on) __builtin_amdgcn_s_setreg((0 << 11) | (23 << 6) | 1, 1);      // hwreg(HW_REG_MODE, 23, 1) = 1
    const float a = in[2 * threadIdx.x], b = in[2 * threadIdx.x + 1];
    f16x2 h = {(_Float16)a, (_Float16)b};
    out[threadIdx.x] = __builtin_bit_cast(unsigned, h);
}
int main() {
    const float vals[16] = {1.0f, -2.5f, 65504.f, 65519.9f, 65520.f, 1e6f, -1e6f, 3e38f, -65520.f, 70000.f, 1e-8f, 6e-5f, __builtin_inff(), -__builtin_inff(), 65503.9f, 0.f};
    float* din; unsigned* dout; unsigned h[8];
    (void)hipMalloc(&din, sizeof(vals)); (void)hipMalloc(&dout, sizeof(h));
    (void)hipMemcpy(din, vals, sizeof(vals), hipMemcpyHostToDevice);
    for (int on = 0; on < 2; ++on) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(8), 0, 0, din, dout, on);
        (void)hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
        printf("FP16_OVFL=%d:", on);
        for (int i = 0; i < 8; ++i) printf(" %04x %04x", h[i] & 0xffff, h[i] >> 16);
        printf("\n");
    }
    printf("(7bff = 65504, 7c00 = +inf, fbff = -65504, fc00 = -inf)\n");
    return 0;
}
