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

// MODE 0 reads + the real store volume: every thread owns a 4x4 patch (TW x TH tile, 256 threads => TW*TH = 4096) and writes its
// four float4 rows of mip 0 (ST = 0: plain stores, 1: nontemporal), plus the 2x2 of mip 1 when M1.  RD = 0 skips the reads.
template <int TW, int TH, int ST, int M1, int RD>
__global__ __launch_bounds__(256) void k_full(const float* __restrict__ img, float* __restrict__ mip0, float* __restrict__ mip1) {
  static_assert(TW * TH == 4096, "4x4 per thread");
  constexpr int TXN = TW / 4;
  const uint32_t tiles_x = 4096 / TW;
  const uint32_t bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
  const uint32_t tx = threadIdx.x % TXN, ty = threadIdx.x / TXN;
  const uint32_t x0 = bx * TW + tx * 4, y0 = by * TH + ty * 4;
  float v[4][4];
#pragma unroll
  for (int r = 0; r < 4; r++)
#pragma unroll
    for (int c = 0; c < 4; c++) {
      uint32_t sx = min(2 * (x0 + c) + 2, DW - 1), sy = min(2 * (y0 + r) + 2, DH - 1);
      v[r][c] = RD ? img[(size_t)sy * DW + sx] : (float)(sx ^ sy);
    }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    float4 o = make_float4(v[r][0], v[r][1], v[r][2], v[r][3]);
    float4* d = reinterpret_cast<float4*>(mip0 + (size_t)(y0 + r) * 4096 + x0);
    typedef float f4v __attribute__((ext_vector_type(4)));
    if (ST)
      __builtin_nontemporal_store(f4v{o.x, o.y, o.z, o.w}, reinterpret_cast<f4v*>(d));
    else
      *d = o;
  }
  if (M1) {
#pragma unroll
    for (int r = 0; r < 2; r++) {
      float2 o = make_float2(fminf(fminf(v[2 * r][0], v[2 * r][1]), fminf(v[2 * r + 1][0], v[2 * r + 1][1])),
                             fminf(fminf(v[2 * r][2], v[2 * r][3]), fminf(v[2 * r + 1][2], v[2 * r + 1][3])));
      float2* d = reinterpret_cast<float2*>(mip1 + (size_t)((y0 >> 1) + r) * 2048 + (x0 >> 1));
      typedef float f2v __attribute__((ext_vector_type(2)));
      if (ST)
        __builtin_nontemporal_store(f2v{o.x, o.y}, reinterpret_cast<f2v*>(d));
      else
        *d = o;
    }
  }
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
  {
    float* mip1;
    hipMalloc(&mip1, (size_t)2048 * 2048 * 4);
    // rotating outputs: a pyramid that is rewritten every 40 us stays in the 256 MB Infinity Cache and its stores never reach HBM;
    // in a frame it was last touched ~0.5 ms and ~2 GB of other traffic ago
    std::vector<float*> outs(NIMG), mip1s(NIMG);
    for (auto& p : outs) hipMalloc(&p, (size_t)4096 * 4096 * 4);
    for (auto& p : mip1s) hipMalloc(&p, (size_t)2048 * 2048 * 4);
#define FULLR(TW, TH, ST, M1, RD, NAME) \
  rep2r(NAME, time_us([&](int i) { hipLaunchKernelGGL((k_full<TW, TH, ST, M1, RD>), dim3(4096), dim3(256), 0, 0, imgs[i % NIMG], outs[i % NIMG], mip1s[i % NIMG]); }, 24));
    auto rep2r = [&](const char* nm, float us) { printf("%-58s %8.1f us  (rotating outputs)\n", nm, us); };
    for (int rep = 0; rep < 2; rep++) {
      FULLR(64, 64, 0, 1, 1, "full: 64x64 read + mip0 + mip1 stores")
      FULLR(64, 64, 1, 1, 1, "full: 64x64 read + mip0 + mip1 nt stores")
      FULLR(64, 64, 0, 1, 0, "full: 64x64 NO read, mip0 + mip1 stores")
      FULLR(64, 64, 1, 1, 0, "full: 64x64 NO read, mip0 + mip1 nt stores")
      FULLR(256, 16, 0, 1, 1, "full: 256x16 read + mip0 + mip1 stores")
      FULLR(1024, 4, 0, 1, 1, "full: 1024x4 read + mip0 + mip1 stores")
      FULLR(1024, 4, 1, 1, 1, "full: 1024x4 read + mip0 + mip1 nt stores")
    }
    auto rep2 = [&](const char* nm, float us) { printf("%-58s %8.1f us\n", nm, us); };
#define FULL(TW, TH, ST, M1, RD, NAME) \
  rep2(NAME, time_us([&](int i) { hipLaunchKernelGGL((k_full<TW, TH, ST, M1, RD>), dim3(4096), dim3(256), 0, 0, imgs[i % NIMG], out, mip1); }, 24));
    for (int rep = 0; rep < 2; rep++) {
      FULL(64, 64, 0, 0, 1, "full: 64x64 read + mip0 stores")
      FULL(64, 64, 1, 0, 1, "full: 64x64 read + mip0 nt stores")
      FULL(64, 64, 0, 1, 1, "full: 64x64 read + mip0 + mip1 stores")
      FULL(64, 64, 0, 1, 0, "full: 64x64 NO read, mip0 + mip1 stores")
      FULL(64, 64, 1, 1, 0, "full: 64x64 NO read, mip0 + mip1 nt stores")
      FULL(128, 32, 0, 1, 1, "full: 128x32 read + mip0 + mip1 stores")
      FULL(256, 16, 0, 1, 1, "full: 256x16 read + mip0 + mip1 stores")
      FULL(256, 16, 1, 1, 1, "full: 256x16 read + mip0 + mip1 nt stores")
      FULL(1024, 4, 0, 1, 1, "full: 1024x4 read + mip0 + mip1 stores")
      FULL(1024, 4, 0, 1, 0, "full: 1024x4 NO read, mip0 + mip1 stores")
      FULL(1024, 4, 1, 1, 1, "full: 1024x4 read + mip0 + mip1 nt stores")
    }
  }
  // with the mip-0 sized store stream next to it (acc per thread: 1/16..1/4 of the real store volume, shape only)
  report("64x64 tile, 4-byte loads + 1 store/thread", time_us([&](int i) { hipLaunchKernelGGL((k_probe<64, 64, 0>), dim3(64 * 64), dim3(256), 0, 0, imgs[i % NIMG], out, sink); }, 24));
  return 0;
}
