// oxcull_hpb.hip -- SURVEY 8(f)-3: producer of the VSM hierarchical page buffer (gfx950).
//
// Replaces the "vsm downsample hpb" pass of Oxylus/src/Render/Passes/Shadowmaps.cpp:331-366 (one dispatch of
// rmvsm_downsample_hpb per mip with an image barrier in between, passes/rmvsm_downsample_hpb.slang:10-33):
// the whole pyramid of a layer is built by ONE block -- 64x64 pages x 10 clipmaps x 7 mips are 55 KB, the
// reference's seven dispatches are pure launch latency -- with a block barrier between levels (a block's own
// global writes are visible to it after __syncthreads()).  Bytes / integer logic only: bit-exact by nature.
#include <hip/hip_runtime.h>

#include "oxcull_device.hpp"
#include "oxcull_kernels.hpp"

namespace oxc {

struct HpbArgs {
  const uint32_t* page_table;  // [layers][h][w] R32UI page metadata
  uint8_t* data;
  uint32_t w, h, layers, levels;
  uint32_t level_off[13];
};

__global__ __launch_bounds__(256) void k_generate_hpb(HpbArgs a) {
  const uint32_t z = blockIdx.x;
  // level 0 (IS_FIRST_PASS): Visible (1) && Backed (4) && Dirty (2), rmvsm.slang:16-28,49-70
  {
    const uint32_t n = a.w * a.h;
    const uint32_t* src = a.page_table + (size_t)z * n;
    uint8_t* dst = a.data + a.level_off[0] + (size_t)z * n;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = (uint8_t)((src[i] & 7u) == 7u);
  }
  for (uint32_t lvl = 1; lvl < a.levels; lvl++) {
    __syncthreads();
    const uint32_t sw = mip_dim(a.w, lvl - 1), sh = mip_dim(a.h, lvl - 1);
    const uint32_t w = mip_dim(a.w, lvl), h = mip_dim(a.h, lvl);  // Shadowmaps.cpp:342-346
    const uint8_t* src = a.data + a.level_off[lvl - 1] + (size_t)z * sw * sh;
    uint8_t* dst = a.data + a.level_off[lvl] + (size_t)z * w * h;
    for (uint32_t i = threadIdx.x; i < w * h; i += blockDim.x) {
      const uint32_t x = i % w, y = i / w;
      uint32_t acc = 0;  // tl | tr | bl | br; texels outside the source level read as 0
#pragma unroll
      for (uint32_t dy = 0; dy < 2; dy++)
#pragma unroll
        for (uint32_t dx = 0; dx < 2; dx++) {
          const uint32_t sx = x * 2 + dx, sy = y * 2 + dy;
          if (sx < sw && sy < sh) acc |= src[sy * sw + sx];
        }
      dst[i] = (uint8_t)(acc == 1u);
    }
  }
}

void launch_generate_hpb(const uint32_t* page_table, uint8_t* data, uint32_t w, uint32_t h, uint32_t layers, uint32_t levels, const uint64_t* level_offset,
                         hipStream_t s) {
  HpbArgs a;
  a.page_table = page_table;
  a.data = data;
  a.w = w;
  a.h = h;
  a.layers = layers;
  a.levels = levels;
  for (uint32_t k = 0; k < 13; k++) a.level_off[k] = k < levels ? (uint32_t)level_offset[k] : 0u;
  if (layers) hipLaunchKernelGGL(k_generate_hpb, dim3(layers), dim3(256), 0, s, a);
}

}  // namespace oxc
