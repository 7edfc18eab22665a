// tools/hiz_probe.hip -- which access shape reads "every other texel of every other row of an 8192^2 float image" fastest when the
// image is NOT cache-resident?  (tools/bw_probe.hip's "even rows" rows re-read one 268 MB image whose touched half fits the 256 MB
// Infinity Cache: they measured the cache.)  Here NIMG images are rotated so every launch streams from HBM.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/hiz_probe.hip -o /tmp/hiz_probe && /tmp/hiz_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

constexpr uint32_t DW = 8192, DH = 8192, NIMG = 6;

// tile = TW mip-0 texels wide x TH tall per 256-thread block; each thread owns (TW*TH/256) texels as RX x RY patch.
// MODE 0: one 4-byte load per texel (depth[2y+2][2x+2]);  MODE 1: 16-byte loads covering the row span (reads both columns).
template <int TW, int TH, int MODE>
__global__ __launch_bounds__(256) void k_probe(const float* __restrict__ img, float* __restrict__ out, uint32_t* sink) {
  constexpr int PER = TW * TH / 256;          // texels per thread
  constexpr int RX = PER >= 4 ? 4 : PER;      // patch width
  constexpr int RY = PER / RX;                // patch height
  constexpr int TXN = TW / RX;                // threads across
  const uint32_t tiles_x = 4096 / TW;
  const uint32_t bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
  const uint32_t tx = threadIdx.x % TXN, ty = threadIdx.x / TXN;
  const uint32_t x0 = bx * TW + tx * RX, y0 = by * TH + ty * RY;
  float acc = 0.f;
  if (MODE == 0) {
    float v[RY][RX];
#pragma unroll
    for (int r = 0; r < RY; r++)
#pragma unroll
      for (int c = 0; c < RX; c++) {
        uint32_t sx = min(2 * (x0 + c) + 2, DW - 1), sy = min(2 * (y0 + r) + 2, DH - 1);
        v[r][c] = img[(size_t)sy * DW + sx];
      }
#pragma unroll
    for (int r = 0; r < RY; r++)
#pragma unroll
      for (int c = 0; c < RX; c++) acc += v[r][c];
  } else {
    // RX = 4 texels -> 8 floats = two aligned float4 (columns 2x0 .. 2x0+7; the +2 shift is ignored: same traffic shape)
    float4 v[RY][2];
#pragma unroll
    for (int r = 0; r < RY; r++) {
      uint32_t sy = min(2 * (y0 + r) + 2, DH - 1);
      const float4* p = reinterpret_cast<const float4*>(img + (size_t)sy * DW + 2 * x0);
      v[r][0] = p[0];
      v[r][1] = p[1];
    }
#pragma unroll
    for (int r = 0; r < RY; r++) acc += v[r][0].x + v[r][0].z + v[r][1].x + v[r][1].z;
  }
  if (out) out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
  if (acc == 1.2345e-30f) *sink = 1;
}

template <class F>
static float time_us(F f, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  f(0);
  hipDeviceSynchronize();
  hipEventRecord(a, 0);
  for (int i = 0; i < reps; i++) f(i + 1);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / reps;
}

int main() {
  std::vector<float*> imgs(NIMG);
  for (auto& p : imgs) {
    hipMalloc(&p, (size_t)DW * DH * 4);
    hipMemset(p, 0x3c, (size_t)DW * DH * 4);
  }
  uint32_t* sink;
  hipMalloc(&sink, 4);
  float* out;
  hipMalloc(&out, (size_t)4096 * 4096 * 4);
  const double useful = 4.0 * 4096 * 4096, lines = 4.0 * 8192 * 4096;
  auto report = [&](const char* nm, float us) { printf("%-58s %8.1f us  %6.0f GB/s useful  %6.0f GB/s line-granular\n", nm, us, useful / us / 1e3, lines / us / 1e3); };
#define RUN(TW, TH, MODE, NAME)                                                                                                            \
  report(NAME, time_us([&](int i) { hipLaunchKernelGGL((k_probe<TW, TH, MODE>), dim3((4096 / TW) * (4096 / TH)), dim3(256), 0, 0, imgs[i % NIMG], (float*)nullptr, sink); }, 24));
  for (int rep = 0; rep < 2; rep++) {
    RUN(64, 64, 0, "64x64 tile, 4-byte loads (k_hiz_tile's shape)")
    RUN(64, 64, 1, "64x64 tile, 16-byte loads")
    RUN(256, 16, 0, "256x16 tile, 4-byte loads")
    RUN(256, 16, 1, "256x16 tile, 16-byte loads")
    RUN(1024, 4, 0, "1024x4 tile, 4-byte loads")
    RUN(1024, 4, 1, "1024x4 tile, 16-byte loads")
    RUN(4096, 1, 1, "4096x1 tile (whole row), 16-byte loads")
    RUN(128, 32, 1, "128x32 tile, 16-byte loads")
  }
  // with the mip-0 sized store stream next to it (acc per thread: 1/16..1/4 of the real store volume, shape only)
  report("64x64 tile, 4-byte loads + 1 store/thread", time_us([&](int i) { hipLaunchKernelGGL((k_probe<64, 64, 0>), dim3(64 * 64), dim3(256), 0, 0, imgs[i % NIMG], out, sink); }, 24));
  return 0;
}
