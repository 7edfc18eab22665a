// tools/cvt_check.hip -- do v_cvt_u32_f32 / v_cvt_i32_f32 / v_cvt_flr_i32_f32 on this GPU have the saturating, NaN -> 0 semantics the
// canonical arithmetic spells out (oxcull_device.hpp cvt_u32_sat / cvt_i32_sat)?  Sweeps special values and 2^26 bit patterns.
//   hipcc --offload-arch=gfx950 -O2 tools/cvt_check.hip -o tools/bin/cvt_check && tools/bin/cvt_check
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
__device__ uint32_t ref_u32(float f) {
  if (!(f > 0.0f)) return 0u;
  if (f >= 4294967296.0f) return 0xFFFFFFFFu;
  return (uint32_t)f;
}
__device__ int32_t ref_i32(float f) {
  if (f != f) return 0;
  if (f >= 2147483648.0f) return 2147483647;
  if (f <= -2147483648.0f) return (int32_t)0x80000000;
  return (int32_t)f;
}
__global__ void k(uint32_t stride, uint32_t n, unsigned long long* bad) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    uint32_t bits = i * stride + (i >> 7);  // covers every exponent and sign many times over
    float f;
    memcpy(&f, &bits, 4);
    uint32_t u;
    int32_t a, b;
    asm("v_cvt_u32_f32_e32 %0, %1" : "=v"(u) : "v"(f));
    asm("v_cvt_i32_f32_e32 %0, %1" : "=v"(a) : "v"(f));
    asm("v_cvt_flr_i32_f32_e32 %0, %1" : "=v"(b) : "v"(f));
    if (u != ref_u32(f)) atomicAdd(&bad[0], 1ull);
    if (a != ref_i32(f)) atomicAdd(&bad[1], 1ull);
    if (b != ref_i32(floorf(f))) atomicAdd(&bad[2], 1ull);
  }
}
int main() {
  unsigned long long* bad;
  hipMalloc(&bad, 24);
  hipMemset(bad, 0, 24);
  hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, 64u, 1u << 26, bad);       // stride 64: all 2^32 / 64 patterns
  hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, 0x9E3779B1u, 1u << 26, bad);  // scrambled
  unsigned long long h[3];
  hipMemcpy(h, bad, 24, hipMemcpyDeviceToHost);
  printf("mismatches: v_cvt_u32_f32 %llu  v_cvt_i32_f32 %llu  v_cvt_flr_i32_f32 %llu\n", h[0], h[1], h[2]);
  return (h[0] | h[1] | h[2]) ? 1 : 0;
}
