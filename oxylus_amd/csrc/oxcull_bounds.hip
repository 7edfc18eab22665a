// oxcull_bounds.hip -- SURVEY 8(f)-1: meshlet bounds producer (asset side), gfx950.
//
// Replaces the per-meshlet loop of Oxylus/src/Asset/AssetManager_GLTF.cpp:683-744 and the position
// quantisation of :573-578.  The normal cone is meshoptimizer's (v1.2, a dependency that is not vendored in
// the reference): meshopt_computeMeshletBounds -> meshopt_computeClusterBounds -> computeBoundingSphere,
// restated from its published algorithm; only the outputs the engine stores (cone_axis_s8, cone_cutoff_s8)
// are produced.  Canonical arithmetic as everywhere else: IEEE binary32, no contraction, left to right,
// correctly rounded sqrt / divide -- the CPU checker under oracle/ states the same operations in
// sequential form and the two must agree byte for byte (tests/test_gpu_bounds.py).
//
// Mapping: one wave per meshlet, lane = triangle (passes of 64 for meshlets of up to 256 triangles).
//  * AABB: lexicographic (value, corner index) wave reductions, i.e. exactly the value the sequential
//    `b < a ? b : a` scan keeps, also when +0.0 and -0.0 meet; min-dot: plain wave minimum;
//  * triangle normals: computed per lane, the non-degenerate ones compacted IN TRIANGLE ORDER into an LDS
//    strip (ballot rank), because the bounding-sphere sweep depends on the order of its points;
//  * bounding sphere of the normals: per-axis extrema as a lexicographic (value, index) wave reduction --
//    exactly the "first index that is strictly smaller/larger" the sequential scan keeps -- then the
//    order-dependent growing sweep, executed redundantly by every lane over the LDS strip (wave-uniform).
#include <hip/hip_runtime.h>

#include "oxcull_device.hpp"
#include "oxcull_kernels.hpp"

#pragma clang fp contract(off)

namespace oxc {

// meshoptimizer.h meshopt_quantizeHalf: nearest (ties away from zero in magnitude), results below 2^-14
// flush to zero, >= 65536 to infinity, every NaN to a quiet NaN.
OXC_DEV uint32_t quantize_half(float v) {
  const uint32_t ui = asu(v);
  const int32_t s = (int32_t)((ui >> 16) & 0x8000u);
  const int32_t em = (int32_t)(ui & 0x7fffffffu);
  int32_t h = (em - (112 << 23) + (1 << 12)) >> 13;
  h = (em < (113 << 23)) ? 0 : h;
  h = (em >= (143 << 23)) ? 0x7c00 : h;
  h = (em > (255 << 23)) ? 0x7e00 : h;
  return (uint32_t)(s | h);
}

// meshoptimizer.h meshopt_quantizeSnorm(v, 8)
OXC_DEV int32_t quantize_snorm8(float v) {
  const float round = (v >= 0.0f ? 0.5f : -0.5f);
  v = (v >= -1.0f) ? v : -1.0f;
  v = (v <= 1.0f) ? v : 1.0f;
  return (int32_t)(v * 127.0f + round);
}

OXC_DEV float wave_min_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float w = __shfl_xor(v, o, 64);
    v = (w < v) ? w : v;
  }
  return v;
}
OXC_DEV float wave_max_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float w = __shfl_xor(v, o, 64);
    v = (v < w) ? w : v;
  }
  return v;
}
// lexicographic reductions: smallest value, lowest index among equals / largest value, lowest index among equals
OXC_DEV void wave_argmin(float& v, uint32_t& i) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float w = __shfl_xor(v, o, 64);
    const uint32_t j = __shfl_xor(i, o, 64);
    const bool take = (w < v) || (w == v && j < i);
    v = take ? w : v;
    i = take ? j : i;
  }
}
OXC_DEV void wave_argmax(float& v, uint32_t& i) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float w = __shfl_xor(v, o, 64);
    const uint32_t j = __shfl_xor(i, o, 64);
    const bool take = (w > v) || (w == v && j < i);
    v = take ? w : v;
    i = take ? j : i;
  }
}

OXC_DEV float dist2(const float* a, const float* b) {
  return ((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1])) + (a[2] - b[2]) * (a[2] - b[2]);
}

constexpr uint32_t kMaxBoundsTris = 256;  // triangles per meshlet the producer accepts (the engine's limit is 64, the wide extension 128)

__global__ __launch_bounds__(256) void k_quantize_positions(const float* __restrict__ pos, uint32_t n, uint2* __restrict__ out) {
  for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) {
    const float x = pos[(size_t)v * 3 + 0], y = pos[(size_t)v * 3 + 1], z = pos[(size_t)v * 3 + 2];
    out[v] = make_uint2(quantize_half(x) | (quantize_half(y) << 16), quantize_half(z));  // u16x4, w = 0 (AssetManager_GLTF.cpp:573-578)
  }
}

__global__ __launch_bounds__(256) void k_meshlet_bounds(const float* __restrict__ pos, const uint4* __restrict__ meshlets, uint32_t meshlet_count,
                                                        const uint32_t* __restrict__ vidx, const uint8_t* __restrict__ micro, uint4* __restrict__ out,
                                                        float* __restrict__ meshlet_minmax) {
  __shared__ float s_normals[4][kMaxBoundsTris][3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float(*normals)[3] = s_normals[wave];
  const float fmax_ = 3.402823466e+38f;
  for (uint32_t m = blockIdx.x * 4 + wave; m < meshlet_count; m += gridDim.x * 4) {
    const uint4 ml = meshlets[m];  // {vertex_offset, tri_offset(bytes), vertex_count, tri_count}
    const uint32_t tcount = min(ml.w, kMaxBoundsTris);
    float bmin[3] = {fmax_, fmax_, fmax_}, bmax[3] = {-fmax_, -fmax_, -fmax_};
    uint32_t bmin_i[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, bmax_i[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    uint32_t triangles = 0;
    for (uint32_t t0 = 0; t0 < ml.w; t0 += 64) {
      const uint32_t t = t0 + (uint32_t)lane;
      float p[3][3];
      bool have = t < ml.w;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        uint32_t vi = 0;
        if (have) vi = vidx[ml.x + micro[ml.y + t * 3u + (uint32_t)k]];
#pragma unroll
        for (int c = 0; c < 3; c++) p[k][c] = have ? pos[(size_t)vi * 3 + c] : 0.0f;
      }
      if (have) {  // AssetManager_GLTF.cpp:690-706 (glm::min(a, b) = b < a ? b : a)
#pragma unroll
        for (int k = 0; k < 3; k++) {
#pragma unroll
          for (int c = 0; c < 3; c++) {
            const uint32_t corner = t * 3u + (uint32_t)k;
            if (p[k][c] < bmin[c]) { bmin[c] = p[k][c]; bmin_i[c] = corner; }
            if (bmax[c] < p[k][c]) { bmax[c] = p[k][c]; bmax_i[c] = corner; }
          }
        }
      }
      // meshopt_computeClusterBounds: triangle normal, degenerate triangles are left out
      const float p10[3] = {p[1][0] - p[0][0], p[1][1] - p[0][1], p[1][2] - p[0][2]};
      const float p20[3] = {p[2][0] - p[0][0], p[2][1] - p[0][1], p[2][2] - p[0][2]};
      const float nx = p10[1] * p20[2] - p10[2] * p20[1];
      const float ny = p10[2] * p20[0] - p10[0] * p20[2];
      const float nz = p10[0] * p20[1] - p10[1] * p20[0];
      const float area = __builtin_sqrtf((nx * nx + ny * ny) + nz * nz);
      const bool valid = have && t < tcount && area != 0.0f;
      const uint64_t vb = __builtin_amdgcn_ballot_w64(valid);
      if (valid) {
        const uint32_t r = triangles + __builtin_amdgcn_mbcnt_hi((uint32_t)(vb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vb, 0u));
        normals[r][0] = nx / area;
        normals[r][1] = ny / area;
        normals[r][2] = nz / area;
      }
      triangles += (uint32_t)__popcll((unsigned long long)vb);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same-wave LDS hand-off
#pragma unroll
    for (int c = 0; c < 3; c++) {
      wave_argmin(bmin[c], bmin_i[c]);
      wave_argmax(bmax[c], bmax_i[c]);
    }
    int32_t axis_s8[3] = {0, 0, 0};
    int32_t cutoff_s8 = 0;  // no valid triangle: cone data stays 0
    if (triangles > 0) {    // wave-uniform
      // ---- computeBoundingSphere(normals, axis_count = 3, radii = 0): per-axis extrema ...
      uint32_t pmin[3], pmax[3];
#pragma unroll
      for (int ax = 0; ax < 3; ax++) {
        float vmin = fmax_, vmax = -fmax_;
        uint32_t imin = 0xFFFFFFFFu, imax = 0xFFFFFFFFu;
        for (uint32_t i = (uint32_t)lane; i < triangles; i += 64) {
          const float tp = normals[i][ax];
          if (tp < vmin) { vmin = tp; imin = i; }
          if (tp > vmax) { vmax = tp; imax = i; }
        }
        wave_argmin(vmin, imin);
        wave_argmax(vmax, imax);
        // the sequential scan starts from index 0 with +-FLT_MAX and only moves on a strict improvement
        pmin[ax] = (vmin < fmax_) ? imin : 0u;
        pmax[ax] = (vmax > -fmax_) ? imax : 0u;
      }
      // ... the most distant pair seeds the sphere ...
      int paxis = 0;
      float paxisdr = 0.0f;
#pragma unroll
      for (int ax = 0; ax < 3; ax++) {
        const float dr = __builtin_sqrtf(dist2(normals[pmax[ax]], normals[pmin[ax]]));
        if (dr > paxisdr) {
          paxisdr = dr;
          paxis = ax;
        }
      }
      const uint32_t i1 = paxis == 0 ? pmin[0] : (paxis == 1 ? pmin[1] : pmin[2]);
      const uint32_t i2 = paxis == 0 ? pmax[0] : (paxis == 1 ? pmax[1] : pmax[2]);
      const float p1[3] = {normals[i1][0], normals[i1][1], normals[i1][2]};
      const float p2[3] = {normals[i2][0], normals[i2][1], normals[i2][2]};
      const float paxisd = __builtin_sqrtf(dist2(p2, p1));
      const float paxisk = paxisd > 0.0f ? paxisd / (2.0f * paxisd) : 0.0f;
      float center[3] = {p1[0] + (p2[0] - p1[0]) * paxisk, p1[1] + (p2[1] - p1[1]) * paxisk, p1[2] + (p2[2] - p1[2]) * paxisk};
      float radius = paxisdr / 2.0f;
      // ... and one order-dependent sweep grows it (every lane runs the same uniform loop)
      for (uint32_t i = 0; i < triangles; i++) {
        const float q[3] = {normals[i][0], normals[i][1], normals[i][2]};
        const float d = __builtin_sqrtf(dist2(q, center));
        if (d > radius) {
          const float k = d > 0.0f ? (d - radius) / (2.0f * d) : 0.0f;
          center[0] += k * (q[0] - center[0]);
          center[1] += k * (q[1] - center[1]);
          center[2] += k * (q[2] - center[2]);
          radius = (radius + d) / 2.0f;
        }
      }
      float axis[3] = {center[0], center[1], center[2]};
      const float axislength = __builtin_sqrtf((axis[0] * axis[0] + axis[1] * axis[1]) + axis[2] * axis[2]);
      const float invaxislength = axislength == 0.0f ? 0.0f : 1.0f / axislength;
      axis[0] *= invaxislength;
      axis[1] *= invaxislength;
      axis[2] *= invaxislength;
      float mindp = 1.0f;
      for (uint32_t i = (uint32_t)lane; i < triangles; i += 64) {
        const float dp = (normals[i][0] * axis[0] + normals[i][1] * axis[1]) + normals[i][2] * axis[2];
        mindp = (dp < mindp) ? dp : mindp;
      }
      mindp = wave_min_f(mindp);
      if (mindp <= 0.1f) {
        cutoff_s8 = 127;  // cone wider than ~168 degrees: never culls; the axis stays 0
      } else {
        const float cone_cutoff = __builtin_sqrtf(1.0f - mindp * mindp);
#pragma unroll
        for (int k = 0; k < 3; k++) axis_s8[k] = quantize_snorm8(axis[k]);
        const float e0 = __builtin_fabsf((float)axis_s8[0] / 127.0f - axis[0]);
        const float e1 = __builtin_fabsf((float)axis_s8[1] / 127.0f - axis[1]);
        const float e2 = __builtin_fabsf((float)axis_s8[2] / 127.0f - axis[2]);
        const int32_t c = (int32_t)(127.0f * (((cone_cutoff + e0) + e1) + e2) + 1.0f);  // rounded up, not to nearest
        cutoff_s8 = c > 127 ? 127 : c;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the strip is rewritten by the next meshlet
    if (lane == 0) {
      uint32_t ch[3], eh[3];
#pragma unroll
      for (int c = 0; c < 3; c++) {  // AssetManager_GLTF.cpp:717-727
        ch[c] = quantize_half((bmax[c] + bmin[c]) * 0.5f);
        eh[c] = quantize_half(bmax[c] - bmin[c]);
      }
      uint4 b;  // GPU::MeshletBounds: center.xyz u16, cone_axis.xy s8, extent.xyz u16, cone_axis.z s8, cutoff s8
      b.x = ch[0] | (ch[1] << 16);
      b.y = ch[2] | (((uint32_t)axis_s8[0] & 0xFFu) << 16) | (((uint32_t)axis_s8[1] & 0xFFu) << 24);
      b.z = eh[0] | (eh[1] << 16);
      b.w = eh[2] | (((uint32_t)axis_s8[2] & 0xFFu) << 16) | (((uint32_t)cutoff_s8 & 0xFFu) << 24);
      out[m] = b;
#pragma unroll
      for (int c = 0; c < 3; c++) {  // reduced in meshlet order by k_mesh_bounds_reduce (AssetManager_GLTF.cpp:736-737)
        meshlet_minmax[(size_t)m * 6 + c] = bmin[c];
        meshlet_minmax[(size_t)m * 6 + 3 + c] = bmax[c];
      }
    }
  }
}

// Mesh AABB = the sequential `b < a ? b : a` fold over the meshlets' boxes in meshlet order
// (AssetManager_GLTF.cpp:736-737,741-744): one block, lexicographic (value, meshlet index) reduction.
__global__ __launch_bounds__(1024) void k_mesh_bounds_reduce(const float* __restrict__ meshlet_minmax, uint32_t meshlet_count, float* __restrict__ out6) {
  __shared__ float s_v[16][6];
  __shared__ uint32_t s_i[16][6];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float fmax_ = 3.402823466e+38f;
  float v[6] = {fmax_, fmax_, fmax_, -fmax_, -fmax_, -fmax_};
  uint32_t idx[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
  for (uint32_t m = threadIdx.x; m < meshlet_count; m += blockDim.x) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float lo = meshlet_minmax[(size_t)m * 6 + c], hi = meshlet_minmax[(size_t)m * 6 + 3 + c];
      if (lo < v[c]) { v[c] = lo; idx[c] = m; }
      if (v[3 + c] < hi) { v[3 + c] = hi; idx[3 + c] = m; }
    }
  }
#pragma unroll
  for (int c = 0; c < 3; c++) {
    wave_argmin(v[c], idx[c]);
    wave_argmax(v[3 + c], idx[3 + c]);
  }
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < 6; c++) {
      s_v[wave][c] = v[c];
      s_i[wave][c] = idx[c];
    }
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int c = 0; c < 6; c++) {
      v[c] = lane < 16 ? s_v[lane][c] : (c < 3 ? fmax_ : -fmax_);
      idx[c] = lane < 16 ? s_i[lane][c] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
      wave_argmin(v[c], idx[c]);
      wave_argmax(v[3 + c], idx[3 + c]);
    }
    if (lane < 3) {
      // (lane-indexed pick without dynamic register indexing)
      const float lo = lane == 0 ? v[0] : (lane == 1 ? v[1] : v[2]);
      const float hi = lane == 0 ? v[3] : (lane == 1 ? v[4] : v[5]);
      out6[lane] = (hi + lo) * 0.5f;
      out6[3 + lane] = hi - lo;
    }
  }
}

void launch_build_meshlet_bounds(const float* pos, uint32_t vertex_count, const void* meshlets, uint32_t meshlet_count, const uint32_t* vidx,
                                 const uint8_t* micro, void* out_bounds, float* out_mesh6, void* out_qpos, float* meshlet_minmax, uint32_t max_grid,
                                 hipStream_t s) {
  if (out_qpos && vertex_count)
    hipLaunchKernelGGL(k_quantize_positions, dim3(min((vertex_count + 255u) / 256u, max_grid)), dim3(256), 0, s, pos, vertex_count,
                       reinterpret_cast<uint2*>(out_qpos));
  if (meshlet_count)
    hipLaunchKernelGGL(k_meshlet_bounds, dim3(min((meshlet_count + 3u) / 4u, max_grid)), dim3(256), 0, s, pos, reinterpret_cast<const uint4*>(meshlets),
                       meshlet_count, vidx, micro, reinterpret_cast<uint4*>(out_bounds), meshlet_minmax);
  hipLaunchKernelGGL(k_mesh_bounds_reduce, dim3(1), dim3(1024), 0, s, meshlet_minmax, meshlet_count, out_mesh6);
}

}  // namespace oxc
