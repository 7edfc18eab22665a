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

// meshoptimizer.h meshopt_quantizeSnorm(v, 10): scale 511, round half away from zero, clamp first (a NaN
// fails both comparisons' keep-branch and becomes -1).
OXC_DEV int32_t quantize_snorm10(float v) {
  const float round = (v >= 0.0f ? 0.5f : -0.5f);
  v = (v >= -1.0f) ? v : -1.0f;
  v = (v <= 1.0f) ? v : 1.0f;
  return (int32_t)(v * 511.0f + round);
}

// AssetManager_GLTF.cpp:578-582: three biased 10-bit fields, x in the top one
__global__ __launch_bounds__(256) void k_quantize_normals(const float* __restrict__ nrm, uint32_t n, uint32_t* __restrict__ out) {
  for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) {
    const float x = nrm[(size_t)v * 3 + 0], y = nrm[(size_t)v * 3 + 1], z = nrm[(size_t)v * 3 + 2];
    out[v] = ((uint32_t)(quantize_snorm10(x) + 511) << 20) | ((uint32_t)(quantize_snorm10(y) + 511) << 10) | (uint32_t)(quantize_snorm10(z) + 511);
  }
}

// AssetManager_GLTF.cpp:585-588
__global__ __launch_bounds__(256) void k_quantize_texcoords(const float2* __restrict__ uv, uint32_t n, uint32_t* __restrict__ out) {
  for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) {
    const float2 t = uv[v];
    out[v] = quantize_half(t.x) | (quantize_half(t.y) << 16);
  }
}

void launch_quantize_vertex_streams(const float* pos, const float* nrm, const float* uv, uint32_t vertex_count, void* out_qpos, void* out_qnrm,
                                    void* out_quv, uint32_t max_grid, hipStream_t s) {
  if (!vertex_count) return;
  const dim3 grid(min((vertex_count + 255u) / 256u, max_grid));
  if (pos) hipLaunchKernelGGL(k_quantize_positions, grid, dim3(256), 0, s, pos, vertex_count, reinterpret_cast<uint2*>(out_qpos));
  if (nrm) hipLaunchKernelGGL(k_quantize_normals, grid, dim3(256), 0, s, nrm, vertex_count, static_cast<uint32_t*>(out_qnrm));
  if (uv) hipLaunchKernelGGL(k_quantize_texcoords, grid, dim3(256), 0, s, reinterpret_cast<const float2*>(uv), vertex_count, static_cast<uint32_t*>(out_quv));
}

// meshopt_computeClusterBounds from the compacted normals on: computeBoundingSphere(normals, radii = 0,
// axis_count = 3) -- per-axis extrema, most distant pair as the seed, one growing sweep --, the axis, the
// minimum dot and the s8 quantisation.  Strictly sequential, exactly as published; `normal_at(i, q)` fetches
// normal i.  Run by ONE lane per meshlet (k_meshlet_cone) or, for oversized meshlets, redundantly by a wave.
template <class F>
OXC_DEV void cone_sequential(F normals8 /* (i0, float q[8][3]): normals i0 .. i0+7, indices clamped to the last one */, uint32_t triangles,
                             int32_t (&axis_s8)[3], int32_t& cutoff_s8) {
  const float fmax_ = 3.402823466e+38f;
  axis_s8[0] = axis_s8[1] = axis_s8[2] = 0;
  cutoff_s8 = 0;  // no valid triangle: cone data stays 0
  if (triangles == 0) return;
  // The three scans below read the normals in blocks of 8 (six 16-byte loads in flight per lane) so that the
  // per-point work does not wait for memory once per point; the order of the points is untouched.
  uint32_t pmin[3] = {0, 0, 0}, pmax[3] = {0, 0, 0};
  float tmin[3] = {fmax_, fmax_, fmax_}, tmax[3] = {-fmax_, -fmax_, -fmax_};
  float pminv[3][3], pmaxv[3][3];  // the extremal points themselves (saves re-fetching them by index)
#pragma unroll
  for (int ax = 0; ax < 3; ax++)
#pragma unroll
    for (int c = 0; c < 3; c++) pminv[ax][c] = pmaxv[ax][c] = 0.0f;
  bool first_point = true;
  for (uint32_t i0 = 0; i0 < triangles; i0 += 8) {
    float q[8][3];
    normals8(i0, q);
    if (first_point) {  // pmin = pmax = 0 initially: the point the sequential code would fetch if nothing ever improves
#pragma unroll
      for (int ax = 0; ax < 3; ax++)
#pragma unroll
        for (int c = 0; c < 3; c++) pminv[ax][c] = pmaxv[ax][c] = q[0][c];
      first_point = false;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t i = i0 + (uint32_t)j;
      if (i < triangles) {
#pragma unroll
        for (int ax = 0; ax < 3; ax++) {
          const float tp = q[j][ax];  // dot with a unit axis: the other products are exact zeros
          if (tp < tmin[ax]) {
            tmin[ax] = tp;
            pmin[ax] = i;
#pragma unroll
            for (int c = 0; c < 3; c++) pminv[ax][c] = q[j][c];
          }
          if (tp > tmax[ax]) {
            tmax[ax] = tp;
            pmax[ax] = i;
#pragma unroll
            for (int c = 0; c < 3; c++) pmaxv[ax][c] = q[j][c];
          }
        }
      }
    }
  }
  float p1[3] = {pminv[0][0], pminv[0][1], pminv[0][2]}, p2[3] = {pmaxv[0][0], pmaxv[0][1], pmaxv[0][2]};  // paxis = 0 unless strictly longer
  float paxisdr = 0.0f;
#pragma unroll
  for (int ax = 0; ax < 3; ax++) {
    const float dr = __builtin_sqrtf(dist2(pmaxv[ax], pminv[ax]));
    if (dr > paxisdr) {
      paxisdr = dr;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        p1[c] = pminv[ax][c];
        p2[c] = pmaxv[ax][c];
      }
    }
  }
  const float paxisd = __builtin_sqrtf(dist2(p2, p1));
  const float paxisk = paxisd > 0.0f ? paxisd / (2.0f * paxisd) : 0.0f;
  float center[3] = {p1[0] + (p2[0] - p1[0]) * paxisk, p1[1] + (p2[1] - p1[1]) * paxisk, p1[2] + (p2[2] - p1[2]) * paxisk};
  float radius = paxisdr / 2.0f;
  for (uint32_t i0 = 0; i0 < triangles; i0 += 8) {
    float q[8][3];
    normals8(i0, q);
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (i0 + (uint32_t)j < triangles) {
        const float d = __builtin_sqrtf(dist2(q[j], center));
        if (d > radius) {
          const float k = d > 0.0f ? (d - radius) / (2.0f * d) : 0.0f;
          center[0] += k * (q[j][0] - center[0]);
          center[1] += k * (q[j][1] - center[1]);
          center[2] += k * (q[j][2] - center[2]);
          radius = (radius + d) / 2.0f;
        }
      }
    }
  }
  float axis[3] = {center[0], center[1], center[2]};
  const float axislength = __builtin_sqrtf((axis[0] * axis[0] + axis[1] * axis[1]) + axis[2] * axis[2]);
  const float invaxislength = axislength == 0.0f ? 0.0f : 1.0f / axislength;
  axis[0] *= invaxislength;
  axis[1] *= invaxislength;
  axis[2] *= invaxislength;
  float mindp = 1.0f;
  for (uint32_t i0 = 0; i0 < triangles; i0 += 8) {
    float q[8][3];
    normals8(i0, q);
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (i0 + (uint32_t)j < triangles) {
        const float dp = (q[j][0] * axis[0] + q[j][1] * axis[1]) + q[j][2] * axis[2];
        mindp = (dp < mindp) ? dp : mindp;
      }
    }
  }
  if (mindp <= 0.1f) {
    cutoff_s8 = 127;  // cone wider than ~168 degrees: never culls; the axis stays 0
    return;
  }
  const float cone_cutoff = __builtin_sqrtf(1.0f - mindp * mindp);
#pragma unroll
  for (int k = 0; k < 3; k++) axis_s8[k] = quantize_snorm8(axis[k]);
  const float e0 = __builtin_fabsf((float)axis_s8[0] / 127.0f - axis[0]);
  const float e1 = __builtin_fabsf((float)axis_s8[1] / 127.0f - axis[1]);
  const float e2 = __builtin_fabsf((float)axis_s8[2] / 127.0f - axis[2]);
  const int32_t c = (int32_t)(127.0f * (((cone_cutoff + e0) + e1) + e2) + 1.0f);  // rounded up, not to nearest
  cutoff_s8 = c > 127 ? 127 : c;
}

// GPU::MeshletBounds: center.xyz u16, cone_axis.xy s8, extent.xyz u16, cone_axis.z s8, cutoff s8 (AssetManager_GLTF.cpp:717-735)
OXC_DEV uint4 pack_bounds(const float* bmin, const float* bmax, const int32_t* axis_s8, int32_t cutoff_s8) {
  uint32_t ch[3], eh[3];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    ch[c] = quantize_half((bmax[c] + bmin[c]) * 0.5f);
    eh[c] = quantize_half(bmax[c] - bmin[c]);
  }
  uint4 b;
  b.x = ch[0] | (ch[1] << 16);
  b.y = ch[2] | (((uint32_t)axis_s8[0] & 0xFFu) << 16) | (((uint32_t)axis_s8[1] & 0xFFu) << 24);
  b.z = eh[0] | (eh[1] << 16);
  b.w = eh[2] | (((uint32_t)axis_s8[2] & 0xFFu) << 16) | (((uint32_t)cutoff_s8 & 0xFFu) << 24);
  return b;
}

constexpr uint32_t kConeDone = 0xFFFFFFFFu;  // normal_counts[] sentinel: the gather kernel already wrote the record

// Kernel 1, one wave per meshlet, lane = triangle: gather the corners, fold the AABB, form the triangle normals
// and write the non-degenerate ones IN TRIANGLE ORDER to normals[m][0..count) (768 B per meshlet, coalesced).
// Meshlets of more than 64 triangles (not produced by the engine) are finished here through an LDS strip.
__global__ __launch_bounds__(256) void k_meshlet_gather(const float* __restrict__ pos, const uint4* __restrict__ meshlets, uint32_t first, uint32_t count,
                                                        const uint32_t* __restrict__ vidx, const uint8_t* __restrict__ micro, uint4* __restrict__ out,
                                                        float* __restrict__ meshlet_minmax, float* __restrict__ normals_out,
                                                        uint32_t* __restrict__ normal_counts) {
  __shared__ float s_normals[4][kMaxBoundsTris][3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float(*lds_normals)[3] = s_normals[wave];
  const float fmax_ = 3.402823466e+38f;
  for (uint32_t k = blockIdx.x * 4 + wave; k < count; k += gridDim.x * 4) {
    const uint32_t m = first + k;
    const uint4 ml = meshlets[m];  // {vertex_offset, tri_offset(bytes), vertex_count, tri_count}
    const uint32_t tcount = min(ml.w, kMaxBoundsTris);
    const bool big = ml.w > 64u;  // wave-uniform
    float* gnormals = normals_out + (size_t)k * 192;
    float bmin[3] = {fmax_, fmax_, fmax_}, bmax[3] = {-fmax_, -fmax_, -fmax_};
    uint32_t bmin_i[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, bmax_i[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    uint32_t triangles = 0;
    for (uint32_t t0 = 0; t0 < ml.w; t0 += 64) {
      const uint32_t t = t0 + (uint32_t)lane;
      float p[3][3];
      const bool have = t < ml.w;
#pragma unroll
      for (int c3 = 0; c3 < 3; c3++) {
        uint32_t vi = 0;
        if (have) vi = vidx[ml.x + micro[ml.y + t * 3u + (uint32_t)c3]];
#pragma unroll
        for (int c = 0; c < 3; c++) p[c3][c] = have ? pos[(size_t)vi * 3 + c] : 0.0f;
      }
      if (have) {  // AssetManager_GLTF.cpp:690-706 (glm::min(a, b) = b < a ? b : a)
#pragma unroll
        for (int c3 = 0; c3 < 3; c3++) {
#pragma unroll
          for (int c = 0; c < 3; c++) {
            const uint32_t corner = t * 3u + (uint32_t)c3;
            if (p[c3][c] < bmin[c]) { bmin[c] = p[c3][c]; bmin_i[c] = corner; }
            if (bmax[c] < p[c3][c]) { bmax[c] = p[c3][c]; bmax_i[c] = corner; }
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
        const float n0 = nx / area, n1 = ny / area, n2 = nz / area;
        if (big) {
          lds_normals[r][0] = n0;
          lds_normals[r][1] = n1;
          lds_normals[r][2] = n2;
        } else {
          typedef float f3 __attribute__((ext_vector_type(3)));
          *reinterpret_cast<f3 __attribute__((aligned(4)))*>(gnormals + r * 3) = f3{n0, n1, n2};  // one 12-byte store
        }
      }
      triangles += (uint32_t)__popcll((unsigned long long)vb);
    }
    // Two distinct bit patterns compare equal only for +0.0 / -0.0, so a plain value reduction already gives
    // the sequential fold's result unless a result is a zero; only then the (value, corner index) form runs.
    {
      float fmin_[3], fmax2_[3];
      bool zero = false;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        fmin_[c] = wave_min_f(bmin[c]);
        fmax2_[c] = wave_max_f(bmax[c]);
        zero = zero || fmin_[c] == 0.0f || fmax2_[c] == 0.0f;
      }
      if (zero) {  // wave-uniform
#pragma unroll
        for (int c = 0; c < 3; c++) {
          wave_argmin(bmin[c], bmin_i[c]);
          wave_argmax(bmax[c], bmax_i[c]);
        }
      } else {
#pragma unroll
        for (int c = 0; c < 3; c++) {
          bmin[c] = fmin_[c];
          bmax[c] = fmax2_[c];
        }
      }
    }
    if (big) {  // finish here: every lane runs the sequential cone code over the LDS strip (wave-uniform)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same-wave LDS hand-off
      int32_t axis_s8[3], cutoff_s8;
      cone_sequential(
          [&](uint32_t i0, float(*q)[3]) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
              const uint32_t i = min(i0 + (uint32_t)j, triangles - 1u);
              q[j][0] = lds_normals[i][0];
              q[j][1] = lds_normals[i][1];
              q[j][2] = lds_normals[i][2];
            }
          },
          triangles, axis_s8, cutoff_s8);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the strip is rewritten by the next meshlet
      if (lane == 0) out[m] = pack_bounds(bmin, bmax, axis_s8, cutoff_s8);
    }
    if (lane == 0) {
      normal_counts[k] = big ? kConeDone : triangles;
#pragma unroll
      for (int c = 0; c < 3; c++) {  // folded in meshlet order by k_mesh_bounds_reduce (AssetManager_GLTF.cpp:736-737)
        meshlet_minmax[(size_t)m * 6 + c] = bmin[c];
        meshlet_minmax[(size_t)m * 6 + 3 + c] = bmax[c];
      }
    }
  }
}

// Kernel 2, one LANE per meshlet: the order-dependent cone code (64 lanes = 64 meshlets instead of 64 copies of one).
__global__ __launch_bounds__(256) void k_meshlet_cone(uint32_t first, uint32_t count, const float* __restrict__ meshlet_minmax,
                                                      const float* __restrict__ normals_in, const uint32_t* __restrict__ normal_counts,
                                                      uint4* __restrict__ out) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  const uint32_t triangles = normal_counts[k];
  if (triangles == kConeDone) return;
  const float* n = normals_in + (size_t)k * 192;
  int32_t axis_s8[3], cutoff_s8;
  cone_sequential(
      [&](uint32_t i0, float(*q)[3]) {  // 8 normals = 96 contiguous bytes = six 16-byte loads; i0 + 7 <= 63 stays inside the 64-normal row
        const float4* src = reinterpret_cast<const float4*>(n + i0 * 3);
        float f[24];
#pragma unroll
        for (int v = 0; v < 6; v++) {
          const float4 t = src[v];
          f[v * 4 + 0] = t.x;
          f[v * 4 + 1] = t.y;
          f[v * 4 + 2] = t.z;
          f[v * 4 + 3] = t.w;
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
          q[j][0] = f[j * 3 + 0];
          q[j][1] = f[j * 3 + 1];
          q[j][2] = f[j * 3 + 2];
        }
      },
      triangles, axis_s8, cutoff_s8);
  const uint32_t m = first + k;
  float bmin[3], bmax[3];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    bmin[c] = meshlet_minmax[(size_t)m * 6 + c];
    bmax[c] = meshlet_minmax[(size_t)m * 6 + 3 + c];
  }
  out[m] = pack_bounds(bmin, bmax, axis_s8, cutoff_s8);
}

// Mesh AABB = the sequential `b < a ? b : a` fold over the meshlets' boxes in meshlet order
// (AssetManager_GLTF.cpp:736-737,741-744) as a lexicographic (value, meshlet index) reduction, which is
// associative: kMeshFoldBlocks blocks fold contiguous ranges into partials {6 values, 6 indices}, one block
// folds the partials and writes centre / extent.
constexpr uint32_t kMeshFoldBlocks = 256;

OXC_DEV void block_fold6(float (&v)[6], uint32_t (&idx)[6], float (*s_v)[6], uint32_t (*s_i)[6]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwaves = (int)(blockDim.x >> 6);
  const float fmax_ = 3.402823466e+38f;
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
      v[c] = lane < nwaves ? s_v[lane][c] : (c < 3 ? fmax_ : -fmax_);
      idx[c] = lane < nwaves ? s_i[lane][c] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
      wave_argmin(v[c], idx[c]);
      wave_argmax(v[3 + c], idx[3 + c]);
    }
  }
}

__global__ __launch_bounds__(256) void k_mesh_bounds_partial(const float* __restrict__ meshlet_minmax, uint32_t meshlet_count, float* __restrict__ part_v,
                                                             uint32_t* __restrict__ part_i) {
  __shared__ float s_v[4][6];
  __shared__ uint32_t s_i[4][6];
  const float fmax_ = 3.402823466e+38f;
  float v[6] = {fmax_, fmax_, fmax_, -fmax_, -fmax_, -fmax_};
  uint32_t idx[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
  for (uint32_t m = blockIdx.x * blockDim.x + threadIdx.x; m < meshlet_count; m += gridDim.x * blockDim.x) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float lo = meshlet_minmax[(size_t)m * 6 + c], hi = meshlet_minmax[(size_t)m * 6 + 3 + c];
      if (lo < v[c]) { v[c] = lo; idx[c] = m; }
      if (v[3 + c] < hi) { v[3 + c] = hi; idx[3 + c] = m; }
    }
  }
  block_fold6(v, idx, s_v, s_i);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int c = 0; c < 6; c++) {
      part_v[blockIdx.x * 6 + c] = v[c];
      part_i[blockIdx.x * 6 + c] = idx[c];
    }
  }
}

__global__ __launch_bounds__(256) void k_mesh_bounds_final(const float* __restrict__ part_v, const uint32_t* __restrict__ part_i, uint32_t parts,
                                                           float* __restrict__ out6) {
  __shared__ float s_v[4][6];
  __shared__ uint32_t s_i[4][6];
  const float fmax_ = 3.402823466e+38f;
  float v[6] = {fmax_, fmax_, fmax_, -fmax_, -fmax_, -fmax_};
  uint32_t idx[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
  if (threadIdx.x < parts) {
#pragma unroll
    for (int c = 0; c < 6; c++) {
      v[c] = part_v[threadIdx.x * 6 + c];
      idx[c] = part_i[threadIdx.x * 6 + c];
    }
  }
  block_fold6(v, idx, s_v, s_i);
  if (threadIdx.x < 3) {
    const int lane = (int)threadIdx.x;  // (lane-indexed pick without dynamic register indexing)
    const float lo = lane == 0 ? v[0] : (lane == 1 ? v[1] : v[2]);
    const float hi = lane == 0 ? v[3] : (lane == 1 ? v[4] : v[5]);
    out6[lane] = (hi + lo) * 0.5f;
    out6[3 + lane] = hi - lo;
  }
}

void launch_build_meshlet_bounds(const float* pos, uint32_t vertex_count, const void* meshlets, uint32_t meshlet_count, const uint32_t* vidx,
                                 const uint8_t* micro, void* out_bounds, float* out_mesh6, void* out_qpos, float* meshlet_minmax, float* normals,
                                 uint32_t* normal_counts, float* fold_scratch /* 256 * 12 words */, uint32_t chunk, uint32_t max_grid, hipStream_t s) {
  if (out_qpos && vertex_count)
    hipLaunchKernelGGL(k_quantize_positions, dim3(min((vertex_count + 255u) / 256u, max_grid)), dim3(256), 0, s, pos, vertex_count,
                       reinterpret_cast<uint2*>(out_qpos));
  // the normals scratch (768 B per meshlet) is bounded by processing `chunk` meshlets at a time
  for (uint32_t first = 0; first < meshlet_count; first += chunk) {
    const uint32_t n = min(chunk, meshlet_count - first);
    hipLaunchKernelGGL(k_meshlet_gather, dim3(min((n + 3u) / 4u, max_grid * 2u)), dim3(256), 0, s, pos, reinterpret_cast<const uint4*>(meshlets), first, n, vidx,
                       micro, reinterpret_cast<uint4*>(out_bounds), meshlet_minmax, normals, normal_counts);
    hipLaunchKernelGGL(k_meshlet_cone, dim3((n + 255u) / 256u), dim3(256), 0, s, first, n, meshlet_minmax, normals, normal_counts,
                       reinterpret_cast<uint4*>(out_bounds));
  }
  // partials live behind the per-meshlet boxes (the caller sized meshlet_minmax for it)
  float* part_v = fold_scratch;
  uint32_t* part_i = reinterpret_cast<uint32_t*>(fold_scratch + kMeshFoldBlocks * 6);
  hipLaunchKernelGGL(k_mesh_bounds_partial, dim3(kMeshFoldBlocks), dim3(256), 0, s, meshlet_minmax, meshlet_count, part_v, part_i);
  hipLaunchKernelGGL(k_mesh_bounds_final, dim3(1), dim3(256), 0, s, part_v, part_i, kMeshFoldBlocks, out_mesh6);
}

}  // namespace oxc
