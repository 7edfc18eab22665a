// tools/microbench.hip -- floor measurements for the 1M-meshlet meshlet-stage launch shape.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o tools/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_empty(unsigned* sink) { if (sink == (unsigned*)1) *sink = 0; }

// read MLI (8 B/lane) + bounds (16 B/lane), G groups per wave, loads batched
template <int G, bool DEVN>
__global__ __launch_bounds__(256) void k_stream(const uint2* __restrict__ mli, const uint4* __restrict__ bnd, const unsigned* __restrict__ nptr, unsigned n_host, unsigned* sink) {
  const unsigned N = DEVN ? nptr[0] : n_host;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned nchunks = (N + 256 * G - 1) / (256 * G);
  unsigned acc = 0;
  for (unsigned chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    uint2 r[G]; uint4 b[G];
#pragma unroll
    for (int j = 0; j < G; j++) { unsigned i = (chunk * 4 * G + j * 4 + wave) * 64 + lane; r[j] = i < N ? mli[i] : make_uint2(0, 0); }
#pragma unroll
    for (int j = 0; j < G; j++) { unsigned i = (chunk * 4 * G + j * 4 + wave) * 64 + lane; b[j] = i < N ? bnd[i] : make_uint4(0, 0, 0, 0); }
#pragma unroll
    for (int j = 0; j < G; j++) acc ^= r[j].x ^ r[j].y ^ b[j].x ^ b[j].y ^ b[j].z ^ b[j].w;
  }
  if (acc == 0x12345678u) *sink = acc;
}

// dependent chain: MLI -> table[mi] (bounds base pointer, L2-resident) -> bounds
template <int G>
__global__ __launch_bounds__(256) void k_chain(const uint2* __restrict__ mli, const unsigned long long* __restrict__ table, const unsigned* __restrict__ nptr, unsigned* sink) {
  const unsigned N = nptr[0];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned nchunks = (N + 256 * G - 1) / (256 * G);
  unsigned acc = 0;
  for (unsigned chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    uint2 r[G]; uint4 b[G]; unsigned long long p[G];
#pragma unroll
    for (int j = 0; j < G; j++) { unsigned i = (chunk * 4 * G + j * 4 + wave) * 64 + lane; r[j] = i < N ? mli[i] : make_uint2(0, 0); }
#pragma unroll
    for (int j = 0; j < G; j++) { unsigned mi = __builtin_amdgcn_readfirstlane(r[j].x); p[j] = table[mi]; }
#pragma unroll
    for (int j = 0; j < G; j++) { b[j] = ((const uint4*)p[j])[r[j].y]; }
#pragma unroll
    for (int j = 0; j < G; j++) acc ^= r[j].x ^ r[j].y ^ b[j].x ^ b[j].y ^ b[j].z ^ b[j].w;
  }
  if (acc == 0x12345678u) *sink = acc;
}

// ALU-only: ~NOPS dependent-ish fp ops per lane per group, no memory
template <int G>
__global__ __launch_bounds__(256) void k_alu(unsigned n, int nops, unsigned* sink) {
  const int lane = threadIdx.x & 63;
  const unsigned nchunks = (n + 256 * G - 1) / (256 * G);
  float a = lane * 0.25f, b = 1.0001f, c = 0.5f, d = 0.125f;
  for (unsigned chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x)
    for (int j = 0; j < G; j++)
      for (int k = 0; k < nops; k += 4) { a = a * b + c; c = c * b + d; d = d * b + a; b = b * 0.9999f + 1e-6f; }
  if (a + c + d == 1234.5f) *sink = 1;
}

int main() {
  const unsigned N = 1000000, M = 1000, K = 1000; const int COPIES = 48;
  std::vector<uint2*> mli(COPIES); std::vector<uint4*> bnd(COPIES); std::vector<unsigned long long*> tab(COPIES);
  unsigned *nptr, *sink; CK(hipMalloc(&nptr, 256)); CK(hipMalloc(&sink, 256)); CK(hipMemcpy(nptr, &N, 4, hipMemcpyHostToDevice));
  std::vector<uint2> h(N); for (unsigned i = 0; i < N; i++) h[i] = make_uint2(i / K, i % K);
  for (int c = 0; c < COPIES; c++) {
    CK(hipMalloc(&mli[c], N * 8)); CK(hipMalloc(&bnd[c], N * 16)); CK(hipMalloc(&tab[c], M * 8));
    CK(hipMemcpy(mli[c], h.data(), N * 8, hipMemcpyHostToDevice)); CK(hipMemset(bnd[c], c + 1, N * 16));
    std::vector<unsigned long long> t(M); for (unsigned m = 0; m < M; m++) t[m] = (unsigned long long)(bnd[c] + (size_t)m * K);
    CK(hipMemcpy(tab[c], t.data(), M * 8, hipMemcpyHostToDevice));
  }
  hipStream_t s; CK(hipStreamCreate(&s)); hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < 96; i++) launch(i % COPIES);
    CK(hipStreamSynchronize(s));
    const int reps = 960;
    CK(hipEventRecord(e0, s)); for (int i = 0; i < reps; i++) launch(i % COPIES); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %8.2f us/launch  (%.0f GB/s of 24 MB)\n", name, ms * 1e3 / reps, 24e6 / (ms * 1e-3 / reps) / 1e9);
  };
  // ramp clocks
  for (int i = 0; i < 20000; i++) hipLaunchKernelGGL((k_stream<2, false>), dim3(2048), dim3(256), 0, s, mli[i % COPIES], bnd[i % COPIES], nptr, N, sink);
  CK(hipStreamSynchronize(s));
  timeit("empty kernel", [&](int) { hipLaunchKernelGGL(k_empty, dim3(2048), dim3(256), 0, s, sink); });
  timeit("stream G=1 grid 4096 hostN", [&](int c) { hipLaunchKernelGGL((k_stream<1, false>), dim3(4096), dim3(256), 0, s, mli[c], bnd[c], nptr, N, sink); });
  timeit("stream G=2 grid 2048 hostN", [&](int c) { hipLaunchKernelGGL((k_stream<2, false>), dim3(2048), dim3(256), 0, s, mli[c], bnd[c], nptr, N, sink); });
  timeit("stream G=2 grid 2048 devN", [&](int c) { hipLaunchKernelGGL((k_stream<2, true>), dim3(2048), dim3(256), 0, s, mli[c], bnd[c], nptr, N, sink); });
  timeit("stream G=4 grid 1024 hostN", [&](int c) { hipLaunchKernelGGL((k_stream<4, false>), dim3(1024), dim3(256), 0, s, mli[c], bnd[c], nptr, N, sink); });
  timeit("stream G=4 grid 1024 devN", [&](int c) { hipLaunchKernelGGL((k_stream<4, true>), dim3(1024), dim3(256), 0, s, mli[c], bnd[c], nptr, N, sink); });
  timeit("stream G=8 grid 512 hostN", [&](int c) { hipLaunchKernelGGL((k_stream<8, false>), dim3(512), dim3(256), 0, s, mli[c], bnd[c], nptr, N, sink); });
  timeit("stream G=4 SAME copy (cache resident)", [&](int) { hipLaunchKernelGGL((k_stream<4, false>), dim3(1024), dim3(256), 0, s, mli[0], bnd[0], nptr, N, sink); });
  timeit("chain G=2 grid 2048 devN", [&](int c) { hipLaunchKernelGGL((k_chain<2>), dim3(2048), dim3(256), 0, s, mli[c], tab[c], nptr, sink); });
  timeit("chain G=4 grid 1024 devN", [&](int c) { hipLaunchKernelGGL((k_chain<4>), dim3(1024), dim3(256), 0, s, mli[c], tab[c], nptr, sink); });
  for (int nops : {100, 200, 300, 400}) {
    char nm[64]; snprintf(nm, 64, "alu only G=4 grid 1024, %d fp ops/lane/group", nops);
    timeit(nm, [&](int) { hipLaunchKernelGGL((k_alu<4>), dim3(1024), dim3(256), 0, s, N, nops, sink); });
  }
  return 0;
}
