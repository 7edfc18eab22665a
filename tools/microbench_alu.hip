// tools/microbench_alu.hip -- achievable VALU rate / shader clock at the meshlet-stage launch shape.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// CH independent fma chains, `iters` rounds -> CH*iters VALU ops per lane
template <int CH, bool PK>
__global__ __launch_bounds__(256) void k_alu(int iters, float* sink, unsigned long long* clk) {
  float v[CH];
  const float b = 1.0001f + threadIdx.x * 1e-9f, c = 0.5f;
#pragma unroll
  for (int k = 0; k < CH; k++) v[k] = threadIdx.x * 0.25f + k;
  unsigned long long t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; i++) {
    if (PK) {
#pragma unroll
      for (int k = 0; k < CH; k += 2) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 x = {v[k], v[k + 1]}, bb = {b, b}, cc = {c, c};
        x = __builtin_elementwise_fma(x, bb, cc);
        v[k] = x.x; v[k + 1] = x.y;
      }
    } else {
#pragma unroll
      for (int k = 0; k < CH; k++) v[k] = __builtin_fmaf(v[k], b, c);
    }
  }
  unsigned long long t1 = clock64(), w1 = wall_clock64();
  float s = 0;
#pragma unroll
  for (int k = 0; k < CH; k++) s += v[k];
  if (s == 1234.5f) *sink = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

int main() {
  float* sink; unsigned long long* clk; CK(hipMalloc(&sink, 256)); CK(hipMalloc(&clk, 256));
  hipStream_t s; CK(hipStreamCreate(&s)); hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto launch, double ops_per_lane, int grid) {
    for (int i = 0; i < 2000; i++) launch();
    CK(hipStreamSynchronize(s));
    const int reps = 500;
    CK(hipEventRecord(e0, s)); for (int i = 0; i < reps; i++) launch(); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    double us = ms * 1e3 / reps;
    double wave_instr = ops_per_lane * grid * 4.0;            // wave-instructions in the launch
    double per_simd = wave_instr / 1024.0;
    printf("%-40s %7.2f us  | %6.0f wave-instr/SIMD -> %5.2f ns/instr/SIMD | clock64/wall(100MHz) = %.2f GHz-equiv\n", name, us, per_simd,
           (us - 2.7) * 1e3 / per_simd, h[1] ? (double)h[0] / h[1] * 0.1 : 0.0);
  };
  const int it = 300;
  run("fma 1 chain , grid 1024 (4 w/SIMD)", [&] { hipLaunchKernelGGL((k_alu<1, false>), dim3(1024), dim3(256), 0, s, it * 4, sink, clk); }, it * 4, 1024);
  run("fma 4 chains, grid 1024 (4 w/SIMD)", [&] { hipLaunchKernelGGL((k_alu<4, false>), dim3(1024), dim3(256), 0, s, it, sink, clk); }, it * 4, 1024);
  run("fma 8 chains, grid 1024 (4 w/SIMD)", [&] { hipLaunchKernelGGL((k_alu<8, false>), dim3(1024), dim3(256), 0, s, it / 2, sink, clk); }, it * 4, 1024);
  run("fma 4 chains, grid 2048 (8 w/SIMD)", [&] { hipLaunchKernelGGL((k_alu<4, false>), dim3(2048), dim3(256), 0, s, it / 2, sink, clk); }, it * 2, 2048);
  run("fma 8 chains, grid 2048 (8 w/SIMD)", [&] { hipLaunchKernelGGL((k_alu<8, false>), dim3(2048), dim3(256), 0, s, it / 4, sink, clk); }, it * 2, 2048);
  run("pk_fma 8 chains, grid 1024 (ops=instr)", [&] { hipLaunchKernelGGL((k_alu<8, true>), dim3(1024), dim3(256), 0, s, it, sink, clk); }, it * 4, 1024);
  run("fma 8 chains, grid 1024, 10x longer", [&] { hipLaunchKernelGGL((k_alu<8, false>), dim3(1024), dim3(256), 0, s, it * 5, sink, clk); }, it * 40, 1024);
  return 0;
}
