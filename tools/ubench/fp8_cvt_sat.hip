// Microbenchmark (diagnostic, not product): does MODE.FP16_OVFL make v_cvt_pk_fp8_f32 saturate on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* in, unsigned* out, int sat) {
  if (sat) __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1);     // hwreg(HW_REG_MODE, 23, 1) = FP16_OVFL
  const float a = in[2 * threadIdx.x], b = in[2 * threadIdx.x + 1];
  out[threadIdx.x] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
}
int main() {
  const float h[16] = {1.f, 447.f, 448.f, 449.f, 464.f, 480.f, 1000.f, 1e30f, -1.f, -448.f, -480.f, -1000.f, -1e30f, 0.f, 2e-10f, 500.f};
  float* d; unsigned* o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 8 * 4);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  for (int sat = 0; sat < 2; ++sat) {
    k<<<1, 8>>>(d, o, sat);
    unsigned r[8]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    printf("FP16_OVFL=%d:", sat);
    for (int i = 0; i < 8; ++i) printf("  %g->0x%02x %g->0x%02x", h[2 * i], r[i] & 0xff, h[2 * i + 1], (r[i] >> 8) & 0xff);
    printf("\n");
  }
  return 0;
}
