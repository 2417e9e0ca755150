// Probe: semantics of gfx950's v_cvt_scalef32_pk_fp8_f32 / v_cvt_scalef32_pk_f32_fp8 (value, scale) -> byte -> value
//   hipcc --offload-arch=gfx950 -O3 tools/scalef32_probe.hip -o tools/diag/scalef32_probe && tools/diag/scalef32_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* x, const float* sc, unsigned* o, float* back) {
    const float a = x[threadIdx.x], s = sc[threadIdx.x];
    s16x2 old = {0, 0};
    const s16x2 w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, a, -a, s, false);
    o[threadIdx.x] = (unsigned)(unsigned short)w[0];
    const f32x2 r = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8((unsigned)(unsigned short)w[0], s, false);
    back[2 * threadIdx.x] = r[0]; back[2 * threadIdx.x + 1] = r[1];
}
int main() {
    const int N = 24;
    float hx[N] = {1.f, 1.f, 1.f, 1.f, 3.f, 0.1f, 17.f, 18.f, 19.f, 20.f, 100.f, 448.f, 480.f, 1000.f, 0.001f, 0.002f, 0.0078125f, 1.0625f, 1.1875f, 1.3125f, 36.f, 44.f, 52.f, 60.f};
    float hs[N] = {1.f, 2.f, 0.5f, 3.9f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    float *dx, *ds, *db; unsigned* dout; unsigned ho[N]; float hb[2 * N];
    (void)hipMalloc(&dx, sizeof(hx)); (void)hipMalloc(&ds, sizeof(hs)); (void)hipMalloc(&dout, sizeof(ho)); (void)hipMalloc(&db, sizeof(hb));
    (void)hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice); (void)hipMemcpy(ds, hs, sizeof(hs), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(N), 0, 0, dx, ds, dout, db);
    (void)hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost); (void)hipMemcpy(hb, db, sizeof(hb), hipMemcpyDeviceToHost);
    for (int i = 0; i < N; ++i) printf("x %10.6f scale %5.2f -> bytes %02x %02x -> back %12.6f %12.6f\n", hx[i], hs[i], ho[i] & 0xff, (ho[i] >> 8) & 0xff, hb[2 * i], hb[2 * i + 1]);
    return 0;
}
