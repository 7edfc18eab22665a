// tools/bw_probe.hip -- what streaming-read rate can this box sustain, and with which access shape?
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/bw_probe.hip -o /tmp/bw_probe && /tmp/bw_probe
// Every variant reads the same 2 GiB (far beyond the 256 MiB Infinity Cache) once per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t fold(u4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

// persistent grid-stride, L independent 16-B loads in flight per lane, tile = L * 4 KiB per block iteration
template <int L, bool NT, bool XCD>
__global__ __launch_bounds__(256) void k_read_persist(const u4* __restrict__ p, uint64_t n16, uint32_t* sink) {
  uint32_t acc = 0;
  const uint64_t tiles = n16 / (256 * L);
  uint64_t first = blockIdx.x, stride = gridDim.x, last = tiles;
  if (XCD) {  // block b runs on XCD b % 8: give every XCD one contiguous eighth of the buffer
    const uint64_t per = tiles / 8;
    first = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    stride = gridDim.x >> 3;
    last = (blockIdx.x & 7) * per + per;
  }
  for (uint64_t t = first; t < last; t += stride) {
    const u4* q = p + t * (256 * L) + threadIdx.x;
    u4 v[L];
#pragma unroll
    for (int i = 0; i < L; i++) v[i] = NT ? __builtin_nontemporal_load(q + i * 256) : q[i * 256];
#pragma unroll
    for (int i = 0; i < L; i++) acc ^= fold(v[i]);
  }
  if (acc == 0x9E3779B9u) *sink = acc;
}

// one tile per block (no loop): grid = tiles
template <int L, bool NT>
__global__ __launch_bounds__(256) void k_read_oneshot(const u4* __restrict__ p, uint32_t* sink) {
  const u4* q = p + (uint64_t)blockIdx.x * (256 * L) + threadIdx.x;
  u4 v[L];
#pragma unroll
  for (int i = 0; i < L; i++) v[i] = NT ? __builtin_nontemporal_load(q + i * 256) : q[i * 256];
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < L; i++) acc ^= fold(v[i]);
  if (acc == 0x9E3779B9u) *sink = acc;
}

// lane-contiguous: each lane reads L consecutive 16-B words (64*L contiguous bytes per lane)
template <int L>
__global__ __launch_bounds__(256) void k_read_lanecontig(const u4* __restrict__ p, uint32_t* sink) {
  const u4* q = p + ((uint64_t)blockIdx.x * 256 + threadIdx.x) * L;
  u4 v[L];
#pragma unroll
  for (int i = 0; i < L; i++) v[i] = q[i];
  uint32_t acc = 0;
#pragma unroll
  for (int i = 0; i < L; i++) acc ^= fold(v[i]);
  if (acc == 0x9E3779B9u) *sink = acc;
}

__global__ __launch_bounds__(256) void k_copy(const u4* __restrict__ p, u4* __restrict__ o, uint64_t n16) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) o[i] = p[i];
}
template <int L>
__global__ __launch_bounds__(256) void k_copy_oneshot(const u4* __restrict__ p, u4* __restrict__ o) {
  const uint64_t base = (uint64_t)blockIdx.x * (256 * L) + threadIdx.x;
  u4 v[L];
#pragma unroll
  for (int i = 0; i < L; i++) v[i] = p[base + i * 256];
#pragma unroll
  for (int i = 0; i < L; i++) o[base + i * 256] = v[i];
}
__global__ __launch_bounds__(256) void k_fill(u4* __restrict__ o, uint64_t n16) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) {
    u4 v = {(uint32_t)i, (uint32_t)(i >> 7), 0x1234567u, (uint32_t)(i * 2654435761u)};
    o[i] = v;
  }
}

// The HiZ build's access shape on an 8192-wide f32 image: only EVEN rows are read.
// TILE = false: a block reads whole rows (grid-stride over even rows); TILE = true: a block reads a 512-byte segment of 64
// even rows (what a 64x64 mip-0 tile touches).  pitch in floats (8192 = tight, 8192 + 64 = padded by 256 B).
template <bool TILE>
__global__ __launch_bounds__(256) void k_read_even_rows(const float* __restrict__ img, uint32_t pitch, uint32_t rows, uint32_t* sink) {
  uint32_t acc = 0;
  if (!TILE) {
    for (uint32_t r = blockIdx.x; r < rows / 2; r += gridDim.x) {
      const u4* row = reinterpret_cast<const u4*>(img + (size_t)(2 * r) * pitch);
      for (uint32_t c = threadIdx.x; c < 8192 / 4; c += 256) acc ^= fold(row[c]);
    }
  } else {
    const uint32_t tiles_x = 8192 / 128, bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;  // 128 floats = 512 B per row
    const uint32_t lane = threadIdx.x & 31, rsub = threadIdx.x >> 5;                             // 32 lanes x 16 B = one segment
    for (uint32_t r = rsub; r < 64; r += 8) {
      const u4* row = reinterpret_cast<const u4*>(img + (size_t)(2 * (by * 64 + r)) * pitch + bx * 128);
      acc ^= fold(row[lane]);
    }
  }
  if (acc == 0x9E3779B9u) *sink = acc;
}

template <class F>
static double time_us(F&& launch, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(); launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; i++) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return ms * 1e3 / reps;
}

int main() {
  const uint64_t bytes = 2ull << 30, n16 = bytes / 16;
  u4 *a, *b; uint32_t* sink;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 64));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, a, n16);
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, b, n16);
  CK(hipDeviceSynchronize());
  // clock ramp
  for (int i = 0; i < 200; i++) hipLaunchKernelGGL((k_read_persist<4, false, false>), dim3(2048), dim3(256), 0, 0, a, n16, sink);
  CK(hipDeviceSynchronize());
  auto report = [&](const char* name, double us, double total_bytes) { printf("%-58s %9.1f us  %7.0f GB/s\n", name, us, total_bytes / (us * 1e-6) / 1e9); };
#define PERSIST(L, NT, XCD, GRID) { char nm[96]; snprintf(nm, 96, "persist L=%d nt=%d xcd=%d grid=%d", L, NT, XCD, GRID); \
    report(nm, time_us([&] { hipLaunchKernelGGL((k_read_persist<L, NT, XCD>), dim3(GRID), dim3(256), 0, 0, a, n16, sink); }, 5), bytes); }
  PERSIST(4, false, false, 2048)
  PERSIST(4, false, false, 1024)
  PERSIST(4, false, false, 4096)
  PERSIST(4, false, false, 8192)
  PERSIST(8, false, false, 2048)
  PERSIST(8, false, false, 1024)
  PERSIST(16, false, false, 1024)
  PERSIST(16, false, false, 512)
  PERSIST(2, false, false, 4096)
  PERSIST(1, false, false, 8192)
  PERSIST(4, true, false, 2048)
  PERSIST(8, true, false, 2048)
  PERSIST(4, false, true, 2048)
  PERSIST(8, false, true, 2048)
  PERSIST(4, true, true, 2048)
#define ONESHOT(L, NT) { char nm[96]; snprintf(nm, 96, "oneshot L=%d nt=%d grid=%llu", L, NT, (unsigned long long)(n16 / (256 * L))); \
    report(nm, time_us([&] { hipLaunchKernelGGL((k_read_oneshot<L, NT>), dim3((uint32_t)(n16 / (256 * L))), dim3(256), 0, 0, a, sink); }, 5), bytes); }
  ONESHOT(1, false)
  ONESHOT(2, false)
  ONESHOT(4, false)
  ONESHOT(8, false)
  ONESHOT(16, false)
  ONESHOT(4, true)
  ONESHOT(8, true)
#define LANEC(L) { char nm[96]; snprintf(nm, 96, "lane-contiguous L=%d (oneshot)", L); \
    report(nm, time_us([&] { hipLaunchKernelGGL((k_read_lanecontig<L>), dim3((uint32_t)(n16 / (256 * L))), dim3(256), 0, 0, a, sink); }, 5), bytes); }
  LANEC(2)
  LANEC(4)
  {
    const uint32_t rows = 8192;  // 8192 x 8192 f32 = 256 MiB (tight); every other row = 128 MiB read per launch
    for (uint32_t pad : {0u, 64u, 32u}) {
      const uint32_t pitch = 8192 + pad;
      char nm[96];
      snprintf(nm, 96, "even rows, whole rows per block, pitch 8192+%u", pad);
      int it = 0;  // rotate through 7 images so that no launch finds its rows in the 256 MB Infinity Cache
      auto image = [&] { return reinterpret_cast<const float*>(a) + (size_t)(it++ % 7) * pitch * rows; };
      report(nm, time_us([&] { hipLaunchKernelGGL((k_read_even_rows<false>), dim3(2048), dim3(256), 0, 0, image(), pitch, rows, sink); }, 21), 4.0 * 8192 * rows / 2);
      snprintf(nm, 96, "even rows, 512 B x 64-row tiles, pitch 8192+%u", pad);
      report(nm, time_us([&] { hipLaunchKernelGGL((k_read_even_rows<true>), dim3(64 * 64), dim3(256), 0, 0, image(), pitch, rows, sink); }, 21), 4.0 * 8192 * rows / 2);
    }
  }
  report("copy grid-stride grid=4096 (read+write bytes)", time_us([&] { hipLaunchKernelGGL(k_copy, dim3(4096), dim3(256), 0, 0, a, b, n16); }, 5), 2.0 * bytes);
  report("copy grid-stride grid=16384 (read+write bytes)", time_us([&] { hipLaunchKernelGGL(k_copy, dim3(16384), dim3(256), 0, 0, a, b, n16); }, 5), 2.0 * bytes);
  report("copy oneshot L=4 (read+write bytes)", time_us([&] { hipLaunchKernelGGL((k_copy_oneshot<4>), dim3((uint32_t)(n16 / 1024)), dim3(256), 0, 0, a, b); }, 5), 2.0 * bytes);
  report("copy oneshot L=1 (read+write bytes)", time_us([&] { hipLaunchKernelGGL((k_copy_oneshot<1>), dim3((uint32_t)(n16 / 256)), dim3(256), 0, 0, a, b); }, 5), 2.0 * bytes);
  report("hipMemcpyAsync D2D (read+write bytes)", time_us([&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); }, 5), 2.0 * bytes);
  report("hipMemsetAsync (write bytes)", time_us([&] { CK(hipMemsetAsync(b, 0, bytes, 0)); }, 5), bytes);
  return 0;
}
