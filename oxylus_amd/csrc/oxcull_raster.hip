// oxcull_raster.hip -- SURVEY 8(f)-2: consumer of cull_geometry's indirect draw (gfx950).
//
// What draw_for_visbuffer does with reordered_indices + the indirect command (Passes/DrawGeometry.cpp:104-190,
// pipeline visbuffer_encode: vs_main passes/visbuffer_encode.slang:24-49, cullMode eBack, depth GreaterOrEqual,
// reversed Z) as a compute rasteriser, so that early cull -> draw -> depth -> generate_hiz -> late cull -> draw can run
// frame after frame without a graphics queue.  The fixed-function rasteriser's exact rules cannot be matched, so
// the rules are stated (include/oxcull.h, oxc_draw_visbuffer) and implemented twice (here and in the CPU checker):
// clipping against w >= 2^-10 and a 64x guard band (round 2: triangles crossing the camera plane used to be dropped), 1/256-pixel snapping, integer edge functions with a top-left rule, z/w interpolated in binary64 from the exact
// edge values, and per pixel the maximum of (depth bits << 32 | vis) through a 64-bit atomic max -- the "R64
// visbuffer" the reference's own note wishes for (visbuffer.slang:43-45), order-independent by construction.
//
// One lane per triangle fetches and sets up; triangles covering at most kSmallSpan x kSmallSpan pixels (almost all
// meshlet triangles) are rasterised by that lane, larger ones go to a list that one block per triangle walks.
#include <hip/hip_runtime.h>

#include "oxcull_device.hpp"
#include "oxcull_kernels.hpp"

#pragma clang fp contract(off)

namespace oxc {

constexpr int64_t kSmallSpan = 8;

struct TriSetup {
  int32_t x[3], y[3];  // 24.8 fixed point, oriented with positive area
  float z[3];
  uint32_t vis;
};

OXC_DEV int64_t edge_fn(int64_t ax, int64_t ay, int64_t bx, int64_t by, int64_t px, int64_t py) { return (bx - ax) * (py - ay) - (by - ay) * (px - ax); }
// top-left rule for positively oriented triangles: an edge owns its pixels when it goes down, or is horizontal going left
OXC_DEV bool edge_inclusive(int64_t ax, int64_t ay, int64_t bx, int64_t by) {
  const int64_t dx = bx - ax, dy = by - ay;
  return dy > 0 || (dy == 0 && dx < 0);
}

static_assert(sizeof(TriSetup) == kTriSetupBytes, "layout");

struct TriRaster {
  int64_t X[3], Y[3];
  float z[3];
  uint32_t vis;
  int64_t area, b0, b1, b2;
  int64_t px0, px1, py0, py1;
  double inv_area;
  int64_t dx0, dx1, dx2, dy0, dy1, dy2;  // change of the three edge functions per pixel step in x / in y (exact)
};

OXC_DEV void tri_prepare(const TriSetup& t, uint32_t W, uint32_t H, TriRaster& r) {
#pragma unroll
  for (int k = 0; k < 3; k++) {
    r.X[k] = t.x[k];
    r.Y[k] = t.y[k];
    r.z[k] = t.z[k];
  }
  r.vis = t.vis;
  r.area = edge_fn(r.X[0], r.Y[0], r.X[1], r.Y[1], r.X[2], r.Y[2]);
  const int64_t minx = min(min(r.X[0], r.X[1]), r.X[2]), maxx = max(max(r.X[0], r.X[1]), r.X[2]);
  const int64_t miny = min(min(r.Y[0], r.Y[1]), r.Y[2]), maxy = max(max(r.Y[0], r.Y[1]), r.Y[2]);
  // pixel (px, py) has its centre at (256 px + 128, 256 py + 128)
  r.px0 = max((minx - 128 + 255) >> 8, (int64_t)0);
  r.py0 = max((miny - 128 + 255) >> 8, (int64_t)0);
  r.px1 = min((maxx - 128) >> 8, (int64_t)W - 1);
  r.py1 = min((maxy - 128) >> 8, (int64_t)H - 1);
  r.b0 = edge_inclusive(r.X[1], r.Y[1], r.X[2], r.Y[2]) ? 0 : -1;
  r.b1 = edge_inclusive(r.X[2], r.Y[2], r.X[0], r.Y[0]) ? 0 : -1;
  r.b2 = edge_inclusive(r.X[0], r.Y[0], r.X[1], r.Y[1]) ? 0 : -1;
  r.inv_area = 1.0 / (double)r.area;  // one reciprocal per triangle
  // E(a->b)(p) = (bx - ax)(py - ay) - (by - ay)(px - ax): one pixel = 256 units
  r.dx0 = -(r.Y[2] - r.Y[1]) * 256;
  r.dy0 = (r.X[2] - r.X[1]) * 256;
  r.dx1 = -(r.Y[0] - r.Y[2]) * 256;
  r.dy1 = (r.X[0] - r.X[2]) * 256;
  r.dx2 = -(r.Y[1] - r.Y[0]) * 256;
  r.dy2 = (r.X[1] - r.X[0]) * 256;
}

// edge values (weights of corners 0, 1, 2) at the centre of pixel (px, py)
OXC_DEV void tri_edges(const TriRaster& r, int64_t px, int64_t py, int64_t& e0, int64_t& e1, int64_t& e2) {
  const int64_t cx = px * 256 + 128, cy = py * 256 + 128;
  e0 = edge_fn(r.X[1], r.Y[1], r.X[2], r.Y[2], cx, cy);
  e1 = edge_fn(r.X[2], r.Y[2], r.X[0], r.Y[0], cx, cy);
  e2 = edge_fn(r.X[0], r.Y[0], r.X[1], r.Y[1], cx, cy);
}
OXC_DEV void tri_fragment(const TriRaster& r, int64_t e0, int64_t e1, int64_t e2, int64_t px, int64_t py, uint32_t W, unsigned long long* visdepth) {
  if (e0 + r.b0 < 0 || e1 + r.b1 < 0 || e2 + r.b2 < 0) return;
  const double zd = (((double)e0 * (double)r.z[0] + (double)e1 * (double)r.z[1]) + (double)e2 * (double)r.z[2]) * r.inv_area;
  const float zf = (float)zd;
  if (!(zf > 0.0f) || zf > 1.0f) return;
  const unsigned long long packed = ((unsigned long long)asu(zf) << 32) | r.vis;
  // (reading the stored value first to skip occluded fragments was measured slower: 1.61 -> 2.15 ms per frame)
  atomicMax(&visdepth[(size_t)py * W + (size_t)px], packed);
}

// Clip planes of the stated rules (include/oxcull.h): w >= kClipWMin and the guard band |x|, |y| <= kClipGuard * w, which keeps every
// screen coordinate inside the +-2^20 px fixed-point range for extents up to 16384.
constexpr float kClipWMin = 0.0009765625f;  // 2^-10
constexpr float kClipGuard = 64.0f;
OXC_DEV float clip_distance(const float* v, int plane) {
  switch (plane) {
    case 0: return v[3] - kClipWMin;
    case 1: return kClipGuard * v[3] - v[0];
    case 2: return kClipGuard * v[3] + v[0];
    case 3: return kClipGuard * v[3] - v[1];
    default: return kClipGuard * v[3] + v[1];
  }
}

// vs_main (visbuffer_encode.slang:24-49) for the three corners of triangle `tri`: clip coordinates + the encoded vis value.
OXC_DEV void tri_clip_coords(const DrawArgs& a, uint32_t tri, float (&clip)[3][4], uint32_t& vis_out) {
  const uint32_t corner_bits = a.wide ? 9u : 8u;
  const uint32_t corner_mask = (1u << corner_bits) - 1u;
  // The three indices of a triangle written by cull_triangles name the same meshlet instance, so everything up to
  // the Meshlet record and the world matrix is fetched once and reused while the instance id repeats (vs_main
  // decodes every index on its own; an index list that mixes instances inside a triangle still works, slower).
  uint32_t cur_mli = 0xFFFFFFFFu;
  uint4 ml = make_uint4(0, 0, 0, 0);
  uint64_t micro = 0, vidx = 0, positions = 0;
  float w[12] = {0};  // rows 0..2 of world
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const uint32_t data = a.indices[tri * 3u + (uint32_t)k];
    const uint32_t mli_index = data >> corner_bits, corner = data & corner_mask;
    if (mli_index != cur_mli) {
      cur_mli = mli_index;
      const uint2 mli = reinterpret_cast<const uint2*>(a.meshlet_instances)[mli_index];
      const GpuMeshInstance inst = a.mesh_instances[mli.x];
      const GpuMesh* mesh = a.meshes + inst.mesh_index;
      const GpuMeshLOD* lod = reinterpret_cast<const GpuMeshLOD*>(mesh->lods) + inst.lod_index;
      ml = load_global_u4(lod->meshlets, mli.y);  // {vertex_offset, tri_offset(bytes), vertex_count, tri_count}
      micro = lod->local_triangle_indices;
      vidx = lod->indirect_vertex_indices;
      positions = mesh->vertex_positions;
      const float* wm = a.transforms + (size_t)inst.transform_index * 16;
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 4; c++) w[r * 4 + c] = OXC_M(wm, r, c);
    }
    const uint32_t boff = ml.y + corner;
    const uint32_t li = (load_global_u32(micro, boff >> 2) >> ((boff & 3u) * 8u)) & 0xFFu;  // scene.slang:336-348
    const uint32_t vi = load_global_u32(vidx, ml.x + li);
    const uint2 q = load_global_u2(positions, vi);  // u16x4
    const float p[3] = {dequantize_half(q.x & 0xFFFFu), dequantize_half(q.x >> 16), dequantize_half(q.y & 0xFFFFu)};
    float world[3];
#pragma unroll
    for (int r = 0; r < 3; r++) world[r] = ((w[r * 4 + 0] * p[0] + w[r * 4 + 1] * p[1]) + w[r * 4 + 2] * p[2]) + w[r * 4 + 3];
#pragma unroll
    for (int r = 0; r < 4; r++) clip[k][r] = ((OXC_M(a.pv, r, 0) * world[0] + OXC_M(a.pv, r, 1) * world[1]) + OXC_M(a.pv, r, 2) * world[2]) + OXC_M(a.pv, r, 3);
    if (k == 0) vis_out = (mli_index << 8) | ((corner / 3u) & 0xFFu);  // VisBufferData(mli, triangle_index / 3).encode()
  }
}

// 0: every corner inside every clip plane (the usual case); 1: crosses a plane (goes to the clipper); 2: all corners outside one plane
OXC_DEV int tri_clip_class(const float (&clip)[3][4]) {
  bool crosses = false;
#pragma unroll
  for (int pl = 0; pl < 5; pl++) {
    const bool i0 = clip_distance(clip[0], pl) >= 0.0f, i1 = clip_distance(clip[1], pl) >= 0.0f, i2 = clip_distance(clip[2], pl) >= 0.0f;
    if (!i0 && !i1 && !i2) return 2;
    crosses |= !(i0 && i1 && i2);
  }
  return crosses ? 1 : 0;
}

// The stated setup rules for one (possibly clipped) triangle given in clip coordinates.  Returns false when it is dropped
// (back face / zero area; w <= 0 or a coordinate beyond the fixed-point range cannot happen behind the clipper but are kept as guards).
OXC_DEV bool tri_finish(const DrawArgs& a, const float* c0, const float* c1, const float* c2, uint32_t vis, TriSetup& out) {
  const float* cl[3] = {c0, c1, c2};
  int64_t X[3], Y[3];
  float z[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float* clip = cl[k];
    if (!(clip[3] > 0.0f)) return false;
    const float sx = ((clip[0] / clip[3]) * 0.5f + 0.5f) * (float)a.width;
    const float sy = ((clip[1] / clip[3]) * 0.5f + 0.5f) * (float)a.height;
    z[k] = clip[2] / clip[3];
    if (!(__builtin_fabsf(sx) <= 1048576.0f) || !(__builtin_fabsf(sy) <= 1048576.0f)) return false;
    X[k] = (int64_t)__builtin_floorf(sx * 256.0f + 0.5f);
    Y[k] = (int64_t)__builtin_floorf(sy * 256.0f + 0.5f);
  }
  out.vis = vis;
  const int64_t area = edge_fn(X[0], Y[0], X[1], Y[1], X[2], Y[2]);
  if (area >= 0) return false;  // cullMode eBack: det(xyw) > 0 <=> positive area; 0 = no coverage
  // orient positively: swap corners 1 and 2
  out.x[0] = (int32_t)X[0];
  out.y[0] = (int32_t)Y[0];
  out.x[1] = (int32_t)X[2];
  out.y[1] = (int32_t)Y[2];
  out.x[2] = (int32_t)X[1];
  out.y[2] = (int32_t)Y[1];
  out.z[0] = z[0];
  out.z[1] = z[2];
  out.z[2] = z[1];
  return true;
}

// rasterise one set-up triangle from this lane (small ones) or queue it for k_draw_big
OXC_DEV void tri_emit(const DrawArgs& a, const TriSetup& t) {
  TriRaster r;
  tri_prepare(t, a.width, a.height, r);
  if (r.px1 < r.px0 || r.py1 < r.py0) return;
  bool small = (r.px1 - r.px0) < kSmallSpan && (r.py1 - r.py0) < kSmallSpan;
  if (!small) {
    const uint32_t slot = atomicAdd(a.big_count, 1u);
    if (slot < a.big_capacity) {
      a.big_list[slot] = t;
      return;
    }
    // the list is full: this lane walks the box itself (slow, correct)
  }
  int64_t r0, r1, r2;  // edge values at the start of the row: stepped exactly (integers) instead of re-multiplied
  tri_edges(r, r.px0, r.py0, r0, r1, r2);
  for (int64_t py = r.py0; py <= r.py1; py++) {
    int64_t e0 = r0, e1 = r1, e2 = r2;
    for (int64_t px = r.px0; px <= r.px1; px++) {
      tri_fragment(r, e0, e1, e2, px, py, a.width, a.visdepth);
      e0 += r.dx0;
      e1 += r.dx1;
      e2 += r.dx2;
    }
    r0 += r.dy0;
    r1 += r.dy1;
    r2 += r.dy2;
  }
}

__global__ __launch_bounds__(256) void k_draw_setup(DrawArgs a) {
  set_half_denorm_flush();
  const uint32_t tris = a.draw_cmd[0] / 3u;  // VkDrawIndexedIndirectCommand.indexCount
  for (uint32_t tri = blockIdx.x * blockDim.x + threadIdx.x; tri < tris; tri += gridDim.x * blockDim.x) {
    float clip[3][4];
    uint32_t vis;
    tri_clip_coords(a, tri, clip, vis);
    const int cls = tri_clip_class(clip);
    if (cls == 2) continue;
    if (cls == 1) {  // rare: crosses the camera plane or the guard band -- clipped by k_draw_clipped, one thread per triangle
      const uint32_t slot = atomicAdd(a.clip_count, 1u);
      if (slot < a.clip_capacity) a.clip_list[slot] = tri;
      continue;
    }
    TriSetup t;
    if (tri_finish(a, clip[0], clip[1], clip[2], vis, t)) tri_emit(a, t);
  }
}

// Sutherland-Hodgman against the five planes, fan triangulation, then the same setup as an unclipped triangle.  A new vertex on a
// crossing edge is always interpolated from its inside end I to its outside end O -- t = d(I) / (d(I) - d(O)), v = I + t (O - I), IEEE
// operations in that order -- so the two triangles that share the edge get the same vertex whatever their winding.
__global__ __launch_bounds__(64) void k_draw_clipped(DrawArgs a) {
  set_half_denorm_flush();
  const uint32_t count = min(*a.clip_count, a.clip_capacity);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    float clip[3][4];
    uint32_t vis;
    tri_clip_coords(a, a.clip_list[i], clip, vis);
    float poly[2][9][4];
    int n = 3, cur = 0;
    for (int k = 0; k < 3; k++)
      for (int c = 0; c < 4; c++) poly[0][k][c] = clip[k][c];
    for (int pl = 0; pl < 5 && n >= 3; pl++) {
      int m = 0;
      for (int k = 0; k < n; k++) {
        const float* p = poly[cur][k];
        const float* q = poly[cur][(k + 1) % n];
        const float dp = clip_distance(p, pl), dq = clip_distance(q, pl);
        const bool ip = dp >= 0.0f, iq = dq >= 0.0f;
        if (ip) {
          for (int c = 0; c < 4; c++) poly[cur ^ 1][m][c] = p[c];
          m++;
        }
        if (ip != iq) {
          const float* I = ip ? p : q;
          const float* O = ip ? q : p;
          const float dI = ip ? dp : dq, dO = ip ? dq : dp;
          const float t = dI / (dI - dO);
          for (int c = 0; c < 4; c++) poly[cur ^ 1][m][c] = I[c] + t * (O[c] - I[c]);
          m++;
        }
      }
      n = m;
      cur ^= 1;
    }
    for (int k = 1; k + 1 < n; k++) {
      TriSetup t;
      if (tri_finish(a, poly[cur][0], poly[cur][k], poly[cur][k + 1], vis, t)) tri_emit(a, t);
    }
  }
}

__global__ __launch_bounds__(256) void k_draw_big(DrawArgs a) {
  const uint32_t count = min(*a.big_count, a.big_capacity);
  for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
    TriRaster r;
    tri_prepare(a.big_list[i], a.width, a.height, r);
    const int64_t bw = r.px1 - r.px0 + 1, bh = r.py1 - r.py0 + 1;
    for (int64_t k = threadIdx.x; k < bw * bh; k += blockDim.x) {
      const int64_t px = r.px0 + k % bw, py = r.py0 + k / bw;
      int64_t e0, e1, e2;
      tri_edges(r, px, py, e0, e1, e2);
      tri_fragment(r, e0, e1, e2, px, py, a.width, a.visdepth);
    }
  }
}

__global__ __launch_bounds__(256) void k_resolve_visbuffer(const unsigned long long* __restrict__ visdepth, uint64_t n, float* __restrict__ depth,
                                                           uint32_t* __restrict__ vis) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const unsigned long long v = visdepth[i];
    if (depth) depth[i] = asf((uint32_t)(v >> 32));
    if (vis) vis[i] = (uint32_t)v;
  }
}

void launch_draw_visbuffer(const DrawArgs& a, bool clear, float* depth_out, uint32_t* vis_out, uint32_t max_grid, hipStream_t s) {
  const uint64_t n = (uint64_t)a.width * a.height;
  if (clear) (void)hipMemsetAsync(a.visdepth, 0, n * 8u, s);
  (void)hipMemsetAsync(a.big_count, 0, 256, s);  // big_count and clip_count live in the same 256-byte header
  hipLaunchKernelGGL(k_draw_setup, dim3(max_grid), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_draw_clipped, dim3(256), dim3(64), 0, s, a);
  hipLaunchKernelGGL(k_draw_big, dim3(max_grid), dim3(256), 0, s, a);
  if (depth_out || vis_out)
    hipLaunchKernelGGL(k_resolve_visbuffer, dim3((uint32_t)std::min<uint64_t>((n + 255) / 256, max_grid)), dim3(256), 0, s, a.visdepth, n, depth_out, vis_out);
}

}  // namespace oxc
