// oxcull_terrain.hip -- SURVEY 8(f)-4: terrain patch cull (gfx950).
//
// Replaces RendererInstance::cull_terrain (Oxylus/src/Render/Passes/Terrain.cpp:159-216) and the pipeline
// terrain_cull (Shaders/passes/terrain_cull.slang:17-83): one lane per patch, the same canonical
// test_frustum / project_aabb / test_occlusion arithmetic as the meshlet path (oxcull_device.hpp), the same
// early / late mask protocol.  One block per 1024 patches; survivors are appended in ascending order -- where the
// reference appends in the order its per-wave atomics happen to land (terrain_cull.slang:68-82) -- the way the meshlet
// stage does it: the test kernel leaves one ballot per wave and one count per block, the emit kernel expands them
// behind the sum of the counts of the blocks before it.  A grid of at most 1024 patches (one block) appends in the
// test kernel itself.  The mask needs no atomics: a wave owns exactly the two mask words of its 64 consecutive patches.
#include <hip/hip_runtime.h>

#include "oxcull_device.hpp"
#include "oxcull_kernels.hpp"

#pragma clang fp contract(off)

namespace oxc {

// FUSED: the whole grid is this one block -- append here instead of publishing ballots for k_cull_terrain_emit.
template <bool FUSED>
__global__ __launch_bounds__(1024) void k_cull_terrain_test(TerrainArgs a) {
  __shared__ uint32_t s_cnt[16];
  __shared__ uint32_t s_level_off[13];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < 13) s_level_off[threadIdx.x] = a.hiz_level_off[threadIdx.x];
  __syncthreads();
  const uint32_t total = a.pcx * a.pcy;
  const bool late = (a.cull_flags & OXC_CULL_LATE_PASS) != 0u;
  const bool occl_or_late = (a.cull_flags & (OXC_CULL_TEST_OCCLUSION | OXC_CULL_LATE_PASS)) != 0u;  // HAS_FLAG(a | b) is "any of"
  const bool frustum = (a.cull_flags & OXC_CULL_TEST_FRUSTUM) != 0u;
  float pl[24];  // the six planes of projection_view (cull.slang:58-71): the same for every patch
  frustum_planes(a.pv, pl);
  HizView hiz;
  hiz.data = a.hiz_data;
  hiz.width = a.hiz_w;
  hiz.height = a.hiz_h;
  hiz.levels = a.hiz_levels;
  hiz.lds = nullptr;
  hiz.lds_off = nullptr;
  hiz.lds_first = a.hiz_levels;  // nothing staged in LDS
  hiz.inv_width = exact_reciprocal_or_zero(a.hiz_w);
  hiz.inv_height = exact_reciprocal_or_zero(a.hiz_h);
  const uint32_t first = blockIdx.x * 1024u;
  const uint32_t patch_index = first + threadIdx.x;
  const bool valid = patch_index < total;
  bool visible = false, emit = false;
  uint32_t word = 0;
  if (valid) {
    const uint32_t px = patch_index % a.pcx, py = patch_index / a.pcx;
    // TerrainData::patch_corner (scene.slang:648-652)
    const float g0x = (float)px / (float)a.pcx, g0y = (float)py / (float)a.pcy;
    const float g1x = (float)(px + 1u) / (float)a.pcx, g1y = (float)(py + 1u) / (float)a.pcy;
    const float cminx = a.world_min[0] + g0x * a.world_size[0], cminy = a.world_min[1] + g0y * a.world_size[1];
    const float cmaxx = a.world_min[0] + g1x * a.world_size[0], cmaxy = a.world_min[1] + g1y * a.world_size[1];
    const float2 b = a.patch_minmax[patch_index];
    const float cx = (cminx + cmaxx) * 0.5f, cy = a.base_height + ((b.x + b.y) * 0.5f) * a.height_scale, cz = (cminy + cmaxy) * 0.5f;
    const float hy = a.height_scale * (b.y - b.x);
    const float ex = cmaxx - cminx, ey = hy > 1e-3f ? hy : 1e-3f, ez = cmaxy - cminy;
    word = a.mask[patch_index >> 5];
    const bool was_visible = ((word >> (patch_index & 31u)) & 1u) != 0u;
    visible = late ? true : was_visible;
    if (frustum) {  // cull.slang:73-83
      const float hx = ex * 0.5f, hyy = ey * 0.5f, hz = ez * 0.5f;
      bool inside = true;
#pragma unroll
      for (int i = 0; i < 6; i++) {
        const float nx = pl[i * 4 + 0], ny = pl[i * 4 + 1], nz = pl[i * 4 + 2], nw = pl[i * 4 + 3];
        const float qx = cx + asf(asu(hx) ^ (asu(nx) & 0x80000000u));
        const float qy = cy + asf(asu(hyy) ^ (asu(ny) & 0x80000000u));
        const float qz = cz + asf(asu(hz) ^ (asu(nz) & 0x80000000u));
        inside = inside && !(dot3(qx, qy, qz, nx, ny, nz) <= -nw);
      }
      visible = visible && inside;
    }
    if (occl_or_late) visible = visible && !aabb_occluded(a.pv, a.near_clip, cx, cy, cz, ex, ey, ez, hiz, s_level_off, visible);
    emit = visible && (!late || !was_visible);
  }
  const uint64_t vbits = __builtin_amdgcn_ballot_w64(visible);
  const uint64_t valid_bits = __builtin_amdgcn_ballot_w64(valid);
  if (occl_or_late && (lane == 0 || lane == 32)) {  // terrain_cull.slang:60-66 without atomics: this wave owns both words
    const uint32_t vb = (uint32_t)(vbits >> lane), ok = (uint32_t)(valid_bits >> lane);
    if (ok) a.mask[(first + wave * 64 + lane) >> 5] = (word & ~ok) | vb;  // (lane 0 / 32 hold the word they read)
  }
  const uint64_t ebits = __builtin_amdgcn_ballot_w64(emit);
  if (lane == 0) s_cnt[wave] = (uint32_t)__popcll((unsigned long long)ebits);
  if (!FUSED && lane == 0) a.emit_bits[blockIdx.x * 16u + wave] = ebits;
  __syncthreads();
  uint32_t before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < 16; w++) {
    const uint32_t c = s_cnt[w];
    before += w < wave ? c : 0u;
    all += c;
  }
  if (FUSED) {
    if (emit) {
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(ebits >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ebits, 0u));
      a.visible[before + rank] = patch_index;
    }
    if (threadIdx.x == 0) {  // VkDrawIndirectCommand (Terrain.cpp:167-169)
      a.draw_cmd[0] = 4;
      a.draw_cmd[1] = all;
      a.draw_cmd[2] = 0;
      a.draw_cmd[3] = 0;
    }
  } else if (threadIdx.x == 0) {
    a.block_counts[blockIdx.x] = all;
  }
}

// Ordered append of the ballots: block b writes behind everything the blocks before it emit.
__global__ __launch_bounds__(1024) void k_cull_terrain_emit(TerrainArgs a) {
  __shared__ uint32_t s_red[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t acc = 0;
  for (uint32_t i = threadIdx.x; i < blockIdx.x; i += 1024u) acc += a.block_counts[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) s_red[wave] = acc;
  const uint64_t ebits = a.emit_bits[blockIdx.x * 16u + wave];
  __syncthreads();
  uint32_t base = 0;
#pragma unroll
  for (int w = 0; w < 16; w++) base += s_red[w];
  __syncthreads();
  if (lane == 0) s_red[wave] = (uint32_t)__popcll((unsigned long long)ebits);
  __syncthreads();
  uint32_t before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < 16; w++) {
    const uint32_t c = s_red[w];
    before += w < wave ? c : 0u;
    all += c;
  }
  if ((ebits >> lane) & 1ull) {
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(ebits >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ebits, 0u));
    a.visible[base + before + rank] = blockIdx.x * 1024u + threadIdx.x;
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {  // VkDrawIndirectCommand (Terrain.cpp:167-169)
    a.draw_cmd[0] = 4;
    a.draw_cmd[1] = base + all;
    a.draw_cmd[2] = 0;
    a.draw_cmd[3] = 0;
  }
}

void launch_cull_terrain(const TerrainArgs& a, hipStream_t s) {
  const uint32_t total = a.pcx * a.pcy, blocks = (total + 1023u) / 1024u;
  if (blocks <= 1u) {
    hipLaunchKernelGGL(k_cull_terrain_test<true>, dim3(1), dim3(1024), 0, s, a);
  } else {
    hipLaunchKernelGGL(k_cull_terrain_test<false>, dim3(blocks), dim3(1024), 0, s, a);
    hipLaunchKernelGGL(k_cull_terrain_emit, dim3(blocks), dim3(1024), 0, s, a);
  }
}

}  // namespace oxc
