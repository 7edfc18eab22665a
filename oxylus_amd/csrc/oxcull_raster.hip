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
// One lane per triangle fetches and sets up.  Triangles whose pixel box is at most kSmallSpan x kSmallSpan (almost all meshlet
// triangles) are rasterised by their wave: the 64 boxes of a wave step are laid end to end (prefix sum of the box sizes) and every
// lane takes one box pixel per iteration, finding its triangle by a binary search over the 64 offsets in LDS -- a lane walking its
// own box made the whole wave wait for the largest box of the 64 (round 1: 0.90 ms for 14.6 M triangles).  Larger
// triangles go to a list that one block per triangle walks.
#include <hip/hip_runtime.h>

#include "oxcull_device.hpp"
#include "oxcull_kernels.hpp"

#pragma clang fp contract(off)

namespace oxc {

constexpr int64_t kSmallSpan = 8;

// inclusive prefix sum across the 64 lanes
OXC_DEV uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}

struct TriSetup {
  int32_t x[3], y[3];  // 24.8 fixed point, oriented with positive area
  float z[3];
  uint32_t vis;
};
constexpr int64_t kBigTile = 64;

OXC_DEV int64_t edge_fn(int64_t ax, int64_t ay, int64_t bx, int64_t by, int64_t px, int64_t py) { return (bx - ax) * (py - ay) - (by - ay) * (px - ax); }
// top-left rule for positively oriented triangles: an edge owns its pixels when it goes down, or is horizontal going left
OXC_DEV bool edge_inclusive(int64_t ax, int64_t ay, int64_t bx, int64_t by) {
  const int64_t dx = bx - ax, dy = by - ay;
  return dy > 0 || (dy == 0 && dx < 0);
}

OXC_DEV bool edge_inclusive32(int32_t ax, int32_t ay, int32_t bx, int32_t by) {  // coordinates are below 2^29 in magnitude: no wrap
  const int32_t dx = bx - ax, dy = by - ay;
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
OXC_DEV void fragment(int64_t e0, int64_t e1, int64_t e2, int64_t b0, int64_t b1, int64_t b2, float z0, float z1, float z2, double inv_area, uint32_t vis,
                      int64_t px, int64_t py, uint32_t W, unsigned long long* visdepth) {
  if (e0 + b0 < 0 || e1 + b1 < 0 || e2 + b2 < 0) return;
  const double zd = (((double)e0 * (double)z0 + (double)e1 * (double)z1) + (double)e2 * (double)z2) * inv_area;
  const float zf = (float)zd;
  if (!(zf > 0.0f) || zf > 1.0f) return;
  const unsigned long long packed = ((unsigned long long)asu(zf) << 32) | vis;
  // (reading the stored value first to skip occluded fragments was measured slower: 1.61 -> 2.15 ms per frame)
  atomicMax(&visdepth[(size_t)py * W + (size_t)px], packed);
}
// The same edge function in 32-bit arithmetic: exact (no wrap) when every coordinate difference that enters it is below 2^15 in
// magnitude.  The 64-bit form is five emulated multi-word operations per edge on this hardware; pixels of the small path (corner
// spread < 2^12) and of most big-list tiles (< 2^15) qualify, and an exact integer is the same integer in either width.
OXC_DEV int32_t edge_fn32(int32_t ax, int32_t ay, int32_t bx, int32_t by, int32_t px, int32_t py) { return (bx - ax) * (py - ay) - (by - ay) * (px - ax); }
OXC_DEV void fragment32(int32_t e0, int32_t e1, int32_t e2, int32_t b0, int32_t b1, int32_t b2, float z0, float z1, float z2, double inv_area, uint32_t vis,
                        uint32_t px, uint32_t py, uint32_t W, unsigned long long* visdepth) {
  if (e0 + b0 < 0 || e1 + b1 < 0 || e2 + b2 < 0) return;
  const double zd = (((double)e0 * (double)z0 + (double)e1 * (double)z1) + (double)e2 * (double)z2) * inv_area;
  const float zf = (float)zd;
  if (!(zf > 0.0f) || zf > 1.0f) return;
  const unsigned long long packed = ((unsigned long long)asu(zf) << 32) | vis;
  atomicMax(&visdepth[(size_t)py * W + (size_t)px], packed);
}
OXC_DEV void tri_fragment(const TriRaster& r, int64_t e0, int64_t e1, int64_t e2, int64_t px, int64_t py, uint32_t W, unsigned long long* visdepth) {
  fragment(e0, e1, e2, r.b0, r.b1, r.b2, r.z[0], r.z[1], r.z[2], r.inv_area, r.vis, px, py, W, visdepth);
}

// What a box pixel needs of its triangle, parked in LDS by the lane that set the triangle up (56 bytes)
struct TriLds {
  int32_t x[3], y[3];
  float z[3];
  uint32_t vis;
  uint32_t inv_area_lo, inv_area_hi;
  uint32_t box;   // px0 | py0 << 16
  uint32_t misc;  // (box width - 1) | edge bias bits << 4 | ceil(256 / box width) << 8
};
constexpr bool small_division_exact() {  // k / bw == (k * ceil(256 / bw)) >> 8 for every box pixel index and width the small path sees
  for (int bw = 1; bw <= (int)kSmallSpan; bw++)
    for (int k = 0; k < (int)(kSmallSpan * kSmallSpan); k++)
      if (k / bw != (k * ((256 + bw - 1) / bw)) >> 8) return false;
  return true;
}
static_assert(small_division_exact(), "box pixel index -> (x, y)");

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

__global__ __launch_bounds__(256) void k_draw_rows(DrawArgs a) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < a.mesh_instance_count; i += gridDim.x * blockDim.x) {
    const GpuMeshInstance inst = a.mesh_instances[i];
    const GpuMesh* mesh = a.meshes + inst.mesh_index;
    const GpuMeshLOD* lod = reinterpret_cast<const GpuMeshLOD*>(mesh->lods) + inst.lod_index;
    DrawRow r;
    r.meshlets = lod->meshlets;
    r.micro = lod->local_triangle_indices;
    r.vidx = lod->indirect_vertex_indices;
    r.positions = mesh->vertex_positions;
    const float* wm = a.transforms + (size_t)inst.transform_index * 16;
#pragma unroll
    for (int rr = 0; rr < 3; rr++)
#pragma unroll
      for (int c = 0; c < 4; c++) r.w[rr * 4 + c] = OXC_M(wm, rr, c);
    a.rows[i] = r;
  }
}

// vs_main (visbuffer_encode.slang:24-49) for the three corners of a triangle given its three index-buffer entries: clip coordinates +
// the encoded vis value.
// first_mli: the MeshletInstance record of idx[0]'s instance when the caller has fetched it already (k_draw_setup, a step ahead).
// An index-buffer entry: the packed u32 of visbuffer.slang:9-14 ((id << 8) | corner; << 9 with wide_triangle_index = 1) or, PAIR
// (wide_triangle_index = 2), the 8-byte pair {u32 meshlet_instance_index, u32 corner}.
template <bool PAIR>
struct IdxEntry {
  uint32_t d;
};
template <>
struct IdxEntry<true> {
  uint32_t id, corner;
};
template <bool PAIR>
OXC_DEV IdxEntry<PAIR> load_entry(const DrawArgs& a, uint32_t i) {
  if constexpr (PAIR) {
    const uint2 v = reinterpret_cast<const uint2*>(a.indices)[i];
    return IdxEntry<true>{v.x, v.y};
  } else {
    return IdxEntry<false>{a.indices[i]};
  }
}
template <bool PAIR>
OXC_DEV uint32_t entry_instance(const DrawArgs& a, const IdxEntry<PAIR>& e) {
  if constexpr (PAIR)
    return e.id;
  else
    return e.d >> (a.wide ? 9u : 8u);
}
template <bool PAIR>
OXC_DEV uint32_t entry_corner(const DrawArgs& a, const IdxEntry<PAIR>& e) {
  if constexpr (PAIR)
    return e.corner;
  else
    return e.d & ((1u << (a.wide ? 9u : 8u)) - 1u);
}
template <bool PAIR>
OXC_DEV void tri_clip_coords(const DrawArgs& a, const IdxEntry<PAIR> (&idx)[3], float (&clip)[3][4], uint32_t& vis_out, const uint2* first_mli = nullptr) {
  // The three indices of a triangle written by cull_triangles name the same meshlet instance, so everything up to
  // the Meshlet record and the world matrix is fetched once and reused while the instance id repeats (vs_main
  // decodes every index on its own; an index list that mixes instances inside a triangle still works, slower).
  uint32_t cur_mli = 0xFFFFFFFFu;
  uint4 ml = make_uint4(0, 0, 0, 0);
  uint64_t micro = 0, vidx = 0, positions = 0;
  float w[12] = {0};  // rows 0..2 of world
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const uint32_t mli_index = entry_instance<PAIR>(a, idx[k]), corner = entry_corner<PAIR>(a, idx[k]);
    if (mli_index != cur_mli) {
      cur_mli = mli_index;
      const uint2 mli = (k == 0 && first_mli) ? *first_mli : reinterpret_cast<const uint2*>(a.meshlet_instances)[mli_index];
      const DrawRow* row = a.rows + mli.x;
      const uint4 p0 = reinterpret_cast<const uint4*>(row)[0], p1 = reinterpret_cast<const uint4*>(row)[1];
      const float4 w0 = reinterpret_cast<const float4*>(row)[2], w1 = reinterpret_cast<const float4*>(row)[3], w2 = reinterpret_cast<const float4*>(row)[4];
      micro = (uint64_t)p0.z | ((uint64_t)p0.w << 32);
      vidx = (uint64_t)p1.x | ((uint64_t)p1.y << 32);
      positions = (uint64_t)p1.z | ((uint64_t)p1.w << 32);
      ml = load_global_u4((uint64_t)p0.x | ((uint64_t)p0.y << 32), mli.y);  // {vertex_offset, tri_offset(bytes), vertex_count, tri_count}
      w[0] = w0.x, w[1] = w0.y, w[2] = w0.z, w[3] = w0.w;
      w[4] = w1.x, w[5] = w1.y, w[6] = w1.z, w[7] = w1.w;
      w[8] = w2.x, w[9] = w2.y, w[10] = w2.z, w[11] = w2.w;
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
// spread_out: the larger of the corners' x and y extents (24.8 units); below 4096 every edge-function value of the triangle's own
// pixel box fits 32 bits with room to spare
OXC_DEV bool tri_finish(const DrawArgs& a, const float* c0, const float* c1, const float* c2, uint32_t vis, TriSetup& out, int32_t* spread_out = nullptr) {
  const float* cl[3] = {c0, c1, c2};
  int32_t X[3], Y[3];  // |s| <= 2^20 pixels: the snapped values fit 32 bits (the stated rules are in integers; 64 bits are only needed for products)
  float z[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float* clip = cl[k];
    if (!(clip[3] > 0.0f)) return false;
    const float sx = ((clip[0] / clip[3]) * 0.5f + 0.5f) * (float)a.width;
    const float sy = ((clip[1] / clip[3]) * 0.5f + 0.5f) * (float)a.height;
    z[k] = clip[2] / clip[3];
    if (!(__builtin_fabsf(sx) <= 1048576.0f) || !(__builtin_fabsf(sy) <= 1048576.0f)) return false;
    X[k] = (int32_t)__builtin_floorf(sx * 256.0f + 0.5f);
    Y[k] = (int32_t)__builtin_floorf(sy * 256.0f + 0.5f);
  }
  out.vis = vis;
  const int32_t spread = max(max(max(X[0], X[1]), X[2]) - min(min(X[0], X[1]), X[2]), max(max(Y[0], Y[1]), Y[2]) - min(min(Y[0], Y[1]), Y[2]));
  if (spread_out) *spread_out = spread;
  bool back;  // cullMode eBack: det(xyw) > 0 <=> positive area; 0 = no coverage
  if (spread < 32768)
    back = edge_fn32(X[0], Y[0], X[1], Y[1], X[2], Y[2]) >= 0;
  else
    back = edge_fn(X[0], Y[0], X[1], Y[1], X[2], Y[2]) >= 0;
  if (back) return false;
  // orient positively: swap corners 1 and 2
  out.x[0] = X[0];
  out.y[0] = Y[0];
  out.x[1] = X[2];
  out.y[1] = Y[2];
  out.x[2] = X[1];
  out.y[2] = Y[1];
  out.z[0] = z[0];
  out.z[1] = z[2];
  out.z[2] = z[1];
  return true;
}

// one lane walks a pixel rectangle of its triangle
OXC_DEV void walk_box(const DrawArgs& a, const TriRaster& r, int64_t x0, int64_t y0, int64_t x1, int64_t y1) {
  int64_t r0, r1, r2;  // edge values at the start of the row: stepped exactly (integers) instead of re-multiplied
  tri_edges(r, x0, y0, r0, r1, r2);
  for (int64_t py = y0; py <= y1; py++) {
    int64_t e0 = r0, e1 = r1, e2 = r2;
    for (int64_t px = x0; px <= x1; px++) {
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

// A triangle whose pixel box exceeds kSmallSpan goes to segment `seg` of the big list (k_draw_big: one wave per triangle).  A triangle
// that does not fit its segment any more is walked by this lane (slow, correct).
OXC_DEV void push_big(const DrawArgs& a, const TriSetup& t, uint32_t seg) {
  const uint32_t slot = atomicAdd(a.big_seg_counts + seg * kBigSegStride, 1u);
  if (slot < a.big_seg_capacity) {
    a.big_list[(size_t)seg * a.big_seg_capacity + slot] = t;
    return;
  }
  TriRaster r;
  tri_prepare(t, a.width, a.height, r);
  for (int64_t py = r.py0; py <= r.py1; py++)
    for (int64_t px = r.px0; px <= r.px1; px++) {
      int64_t e0, e1, e2;
      tri_edges(r, px, py, e0, e1, e2);
      tri_fragment(r, e0, e1, e2, px, py, a.width, a.visdepth);
    }
}

// rasterise one set-up triangle from this lane (small ones) or queue it for k_draw_big
OXC_DEV void tri_emit(const DrawArgs& a, const TriSetup& t, uint32_t seg) {
  TriRaster r;
  tri_prepare(t, a.width, a.height, r);
  if (r.px1 < r.px0 || r.py1 < r.py0) return;
  const bool small = (r.px1 - r.px0) < kSmallSpan && (r.py1 - r.py0) < kSmallSpan;
  if (!small) {
    push_big(a, t, seg);
    return;
  }
  walk_box(a, r, r.px0, r.py0, r.px1, r.py1);
}

template <bool PAIR>
__global__ __launch_bounds__(256, 5) void k_draw_setup(DrawArgs a) {
  set_half_denorm_flush();
  __shared__ TriLds s_tri[4][64];
  __shared__ uint32_t s_off[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  TriLds* const tl = s_tri[wave];
  uint32_t* const off = s_off[wave];
  // VkDrawIndexedIndirectCommand.indexCount (pairs: a cull call whose index count wrapped zeroed instanceCount, the draw is a no-op)
  const uint32_t tris = (PAIR && a.draw_cmd[1] == 0u) ? 0u : a.draw_cmd[0] / 3u;
  const uint32_t wave_id = blockIdx.x * 4u + (uint32_t)wave, nwaves = gridDim.x * 4u;
  // The fetches of a triangle hang on each other (index -> MeshletInstance -> row -> Meshlet record -> micro index -> vertex id ->
  // position) and the kernel waits for them most of its time: the index entries are fetched two steps ahead and the MeshletInstance
  // record of the first corner one step ahead, which takes the first two links out of a step's own chain.
  IdxEntry<PAIR> nidx[3] = {}, nnidx[3] = {};            // the next / next but one step's index-buffer entries
  uint2 nmli = make_uint2(0u, 0u);                       // the next step's MeshletInstance record (of its first corner)
  if (wave_id * 64u + (uint32_t)lane < tris) {
#pragma unroll
    for (int k = 0; k < 3; k++) nidx[k] = load_entry<PAIR>(a, (wave_id * 64u + (uint32_t)lane) * 3u + (uint32_t)k);
    nmli = reinterpret_cast<const uint2*>(a.meshlet_instances)[entry_instance<PAIR>(a, nidx[0])];
  }
  if ((wave_id + nwaves) * 64u + (uint32_t)lane < tris) {
#pragma unroll
    for (int k = 0; k < 3; k++) nnidx[k] = load_entry<PAIR>(a, ((wave_id + nwaves) * 64u + (uint32_t)lane) * 3u + (uint32_t)k);
  }
  for (uint32_t base = wave_id * 64u; base < tris; base += nwaves * 64u) {  // wave-uniform
    const uint32_t tri = base + (uint32_t)lane;
    const IdxEntry<PAIR> idx[3] = {nidx[0], nidx[1], nidx[2]};
    const uint2 mli0 = nmli;
    {
#pragma unroll
      for (int k = 0; k < 3; k++) nidx[k] = nnidx[k];
      if (tri + nwaves * 64u < tris) nmli = reinterpret_cast<const uint2*>(a.meshlet_instances)[entry_instance<PAIR>(a, nidx[0])];
      const uint32_t nt = tri + 2u * nwaves * 64u;
      if (nt < tris) {
#pragma unroll
        for (int k = 0; k < 3; k++) nnidx[k] = load_entry<PAIR>(a, nt * 3u + (uint32_t)k);
      }
    }
    uint32_t count = 0;  // box pixels this lane's triangle contributes to the wave's small-triangle pass
    if (tri < tris) {
      float clip[3][4];
      uint32_t vis;
      tri_clip_coords<PAIR>(a, idx, clip, vis, &mli0);
      const int cls = tri_clip_class(clip);
      TriSetup t;
      if (cls == 1) {  // rare: crosses the camera plane or the guard band -- clipped by k_draw_clipped, one thread per triangle
        const uint32_t slot = atomicAdd(a.clip_count, 1u);
        if (slot < a.clip_capacity) a.clip_list[slot] = tri;
      } else if (int32_t spread = 0; cls == 0 && tri_finish(a, clip[0], clip[1], clip[2], vis, t, &spread)) {
        // pixel box (tri_prepare's, in 32 bits): pixel (px, py) has its centre at (256 px + 128, 256 py + 128)
        const int32_t minx = min(min(t.x[0], t.x[1]), t.x[2]), maxx = max(max(t.x[0], t.x[1]), t.x[2]);
        const int32_t miny = min(min(t.y[0], t.y[1]), t.y[2]), maxy = max(max(t.y[0], t.y[1]), t.y[2]);
        const int32_t px0 = max((minx - 128 + 255) >> 8, 0), py0 = max((miny - 128 + 255) >> 8, 0);
        const int32_t px1 = min((maxx - 128) >> 8, (int32_t)a.width - 1), py1 = min((maxy - 128) >> 8, (int32_t)a.height - 1);
        if (px1 >= px0 && py1 >= py0) {
          const int32_t bw = px1 - px0 + 1, bh = py1 - py0 + 1;
          // (the spread test only fails for triangles reaching far outside the image whose clamped box is small: they take the big path)
          if (bw <= (int32_t)kSmallSpan && bh <= (int32_t)kSmallSpan && spread < 4096) {
            count = (uint32_t)(bw * bh);
            TriLds o;
#pragma unroll
            for (int k = 0; k < 3; k++) {
              o.x[k] = t.x[k];
              o.y[k] = t.y[k];
              o.z[k] = t.z[k];
            }
            o.vis = t.vis;
            const int32_t area = edge_fn32(t.x[0], t.y[0], t.x[1], t.y[1], t.x[2], t.y[2]);  // > 0 (oriented), < 2^25
            const unsigned long long ia = __builtin_bit_cast(unsigned long long, 1.0 / (double)area);  // one reciprocal per triangle (tri_prepare)
            o.inv_area_lo = (uint32_t)ia;
            o.inv_area_hi = (uint32_t)(ia >> 32);
            o.box = (uint32_t)px0 | ((uint32_t)py0 << 16);
            const uint32_t bias = (edge_inclusive32(t.x[1], t.y[1], t.x[2], t.y[2]) ? 0u : 1u) | (edge_inclusive32(t.x[2], t.y[2], t.x[0], t.y[0]) ? 0u : 2u) |
                                  (edge_inclusive32(t.x[0], t.y[0], t.x[1], t.y[1]) ? 0u : 4u);
            o.misc = (uint32_t)(bw - 1) | (bias << 4) | (((256u + (uint32_t)bw - 1u) / (uint32_t)bw) << 8);
            tl[lane] = o;
          } else {
            push_big(a, t, wave_id % kBigSegs);
          }
        }
      }
    }
    // ---- the wave's small triangles, one box pixel per lane and iteration
    const uint32_t incl = wave_incl_scan(count, lane);
    const uint32_t total = readlane_u(incl, 63);
    off[lane] = incl - count;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same-wave LDS hand-off: in order, no barrier needed
    for (uint32_t g0 = 0; g0 < total; g0 += 64u) {
      const uint32_t g = g0 + (uint32_t)lane;
      if (g < total) {
        uint32_t j = 0;  // the last triangle whose first box pixel is at or before g (empty boxes share their successor's offset)
#pragma unroll
        for (uint32_t stp = 32u; stp >= 1u; stp >>= 1)
          if (off[j + stp] <= g) j += stp;
        const uint32_t k = g - off[j];
        const TriLds q = tl[j];
        const uint32_t bw = (q.misc & 7u) + 1u, m = q.misc >> 8;
        const uint32_t qy = (k * m) >> 8, qx = k - qy * bw;
        const uint32_t px = (q.box & 0xFFFFu) + qx, py = (q.box >> 16) + qy;
        const int32_t cx = (int32_t)px * 256 + 128, cy = (int32_t)py * 256 + 128;
        const int32_t e0 = edge_fn32(q.x[1], q.y[1], q.x[2], q.y[2], cx, cy);  // corner spread < 2^12, centre inside the box: |e| < 2^25
        const int32_t e1 = edge_fn32(q.x[2], q.y[2], q.x[0], q.y[0], cx, cy);
        const int32_t e2 = edge_fn32(q.x[0], q.y[0], q.x[1], q.y[1], cx, cy);
        const double inv_area = __builtin_bit_cast(double, (unsigned long long)q.inv_area_lo | ((unsigned long long)q.inv_area_hi << 32));
        fragment32(e0, e1, e2, (q.misc & 0x10u) ? -1 : 0, (q.misc & 0x20u) ? -1 : 0, (q.misc & 0x40u) ? -1 : 0, q.z[0], q.z[1], q.z[2], inv_area, q.vis, px, py,
                   a.width, a.visdepth);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the LDS rows are rewritten by the next step
  }
}

// Sutherland-Hodgman against the five planes, fan triangulation, then the same setup as an unclipped triangle.  A new vertex on a
// crossing edge is always interpolated from its inside end I to its outside end O -- t = d(I) / (d(I) - d(O)), v = I + t (O - I), IEEE
// operations in that order -- so the two triangles that share the edge get the same vertex whatever their winding.
// RESCAN: the overflow pass.  More triangles crossed a clip plane than the id queue holds (clip_capacity): the ids beyond it were
// not recorded, so this instantiation walks the whole index list again and clips every crossing triangle it finds.  Drawing a
// triangle twice leaves the image unchanged (per-pixel maximum), so no bookkeeping of which ones the queue did hold is needed.
// It returns at once when the queue did not overflow.
template <bool RESCAN, bool PAIR>
__global__ __launch_bounds__(64) void k_draw_clipped(DrawArgs a) {
  set_half_denorm_flush();
  if (RESCAN && *a.clip_count <= a.clip_capacity) return;
  const uint32_t count = RESCAN ? ((PAIR && a.draw_cmd[1] == 0u) ? 0u : a.draw_cmd[0] / 3u) : min(*a.clip_count, a.clip_capacity);
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
    float clip[3][4];
    uint32_t vis;
    const uint32_t tri = RESCAN ? i : a.clip_list[i];
    const IdxEntry<PAIR> idx[3] = {load_entry<PAIR>(a, tri * 3u), load_entry<PAIR>(a, tri * 3u + 1u), load_entry<PAIR>(a, tri * 3u + 2u)};
    tri_clip_coords<PAIR>(a, idx, clip, vis);
    if (RESCAN && tri_clip_class(clip) != 1) continue;
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
      if (tri_finish(a, poly[cur][0], poly[cur][k], poly[cur][k + 1], vis, t)) tri_emit(a, t, blockIdx.x % kBigSegs);
    }
  }
}

// A pixel rectangle of a big triangle's box, walked by a wave in 8 x 8 pixel blocks (lane = pixel of the block), 64-bit edge functions.
OXC_DEV void raster_rect64(const DrawArgs& a, const TriRaster& r, int64_t ox, int64_t oy, int64_t x1, int64_t y1, int lane) {
  for (int64_t by = oy; by <= y1; by += 8)
    for (int64_t bx = ox; bx <= x1; bx += 8) {
      const int64_t px = bx + (lane & 7), py = by + (lane >> 3);
      if (px <= x1 && py <= y1) {
        int64_t e0, e1, e2;
        tri_edges(r, px, py, e0, e1, e2);
        tri_fragment(r, e0, e1, e2, px, py, a.width, a.visdepth);
      }
    }
}
// The same with 32-bit edge functions when they are exact for this rectangle (wave-uniform test).
OXC_DEV void raster_rect(const DrawArgs& a, const TriRaster& r, int64_t ox, int64_t oy, int64_t x1, int64_t y1, int lane) {
  // every difference the edge functions of this rectangle see: corner - corner, pixel centre - corner
  const int64_t cx0 = ox * 256 + 128, cx1 = x1 * 256 + 128, cy0 = oy * 256 + 128, cy1 = y1 * 256 + 128;
  const int64_t lox = min(min(min(r.X[0], r.X[1]), r.X[2]), cx0), hix = max(max(max(r.X[0], r.X[1]), r.X[2]), cx1);
  const int64_t loy = min(min(min(r.Y[0], r.Y[1]), r.Y[2]), cy0), hiy = max(max(max(r.Y[0], r.Y[1]), r.Y[2]), cy1);
  if (hix - lox < 32768 && hiy - loy < 32768) {
    const int32_t X0 = (int32_t)r.X[0], X1 = (int32_t)r.X[1], X2 = (int32_t)r.X[2], Y0 = (int32_t)r.Y[0], Y1 = (int32_t)r.Y[1], Y2 = (int32_t)r.Y[2];
    for (int32_t by = (int32_t)oy; by <= (int32_t)y1; by += 8)
      for (int32_t bx = (int32_t)ox; bx <= (int32_t)x1; bx += 8) {
        const int32_t px = bx + (lane & 7), py = by + (lane >> 3);
        if (px <= (int32_t)x1 && py <= (int32_t)y1) {
          const int32_t cx = px * 256 + 128, cy = py * 256 + 128;
          fragment32(edge_fn32(X1, Y1, X2, Y2, cx, cy), edge_fn32(X2, Y2, X0, Y0, cx, cy), edge_fn32(X0, Y0, X1, Y1, cx, cy), (int32_t)r.b0, (int32_t)r.b1,
                     (int32_t)r.b2, r.z[0], r.z[1], r.z[2], r.inv_area, r.vis, (uint32_t)px, (uint32_t)py, a.width, a.visdepth);
        }
      }
  } else {
    raster_rect64(a, r, ox, oy, x1, y1, lane);
  }
}

// One wave per big triangle.  Round 1 gave every big triangle a 256-thread block: 0.43 ms for the 330 K triangles of the loop
// benchmark whose boxes are barely larger than 8 x 8, and a full-screen triangle was a single block's 16 K iterations.  Here a box of
// at most 64 x 64 pixels (nearly all) is walked at once; a larger one is cut into 64 x 64 tiles that go to the tile list
// (k_draw_big_tiles: one wave per tile), so a screen-filling triangle becomes a thousand balanced items.
__global__ __launch_bounds__(256) void k_draw_big(DrawArgs a) {
  __shared__ uint32_t s_prefix[kBigSegs + 1];
  __shared__ uint32_t s_wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  {  // the segments' fill counts -> exclusive prefix (every block computes the same 256-entry scan)
    static_assert(kBigSegs == 256, "one count per thread");
    const uint32_t c = min(a.big_seg_counts[threadIdx.x * kBigSegStride], a.big_seg_capacity);
    const uint32_t incl = wave_incl_scan(c, lane);
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; w++) base += s_wsum[w];
    s_prefix[threadIdx.x + 1] = base + incl;
    if (threadIdx.x == 0) s_prefix[0] = 0;
    __syncthreads();
  }
  const uint32_t total = s_prefix[kBigSegs];
  const uint32_t wave_id = blockIdx.x * 4u + (uint32_t)wave, nwaves = gridDim.x * 4u;
  for (uint32_t i = wave_id; i < total; i += nwaves) {  // wave-uniform
    uint32_t seg = 0;  // the segment task i falls into: the last one whose prefix is <= i
#pragma unroll
    for (uint32_t stp = kBigSegs / 2; stp >= 1u; stp >>= 1)
      if (s_prefix[seg + stp] <= i) seg += stp;
    const uint32_t slot = seg * a.big_seg_capacity + (i - s_prefix[seg]);
    TriRaster r;
    tri_prepare(a.big_list[slot], a.width, a.height, r);
    const uint32_t ntx = (uint32_t)((r.px1 - r.px0 + kBigTile) / kBigTile), nty = (uint32_t)((r.py1 - r.py0 + kBigTile) / kBigTile);
    const uint32_t n = ntx * nty;
    if (n == 1u) {
      raster_rect(a, r, r.px0, r.py0, r.px1, r.py1, lane);
      continue;
    }
    uint32_t first = 0;
    if (lane == 0) first = atomicAdd(a.tile_count, n);
    first = readlane_u(first, 0);
    for (uint32_t k = (uint32_t)lane; k < n; k += 64u) {
      if (first + k < a.tile_capacity && first + k >= first) a.tile_list[first + k] = make_uint2(slot, k);
    }
    if (first + n > a.tile_capacity || first + n < first) {  // the tile list is full: the tiles that did not fit are walked here (slow, correct)
#pragma clang loop unroll(disable)
      for (uint32_t k = 0; k < n; k++) {
        if (first + k < a.tile_capacity && first + k >= first) continue;
        const int64_t ox = r.px0 + (int64_t)(k % ntx) * kBigTile, oy = r.py0 + (int64_t)(k / ntx) * kBigTile;
        raster_rect64(a, r, ox, oy, min(ox + kBigTile - 1, r.px1), min(oy + kBigTile - 1, r.py1), lane);
      }
    }
  }
}

// One wave per tile-list item: a 64 x 64 pixel tile of a triangle whose box is larger than that.
__global__ __launch_bounds__(256) void k_draw_big_tiles(DrawArgs a) {
  const uint32_t count = min(*a.tile_count, a.tile_capacity);
  const int lane = threadIdx.x & 63;
  const uint32_t wave_id = blockIdx.x * 4u + (threadIdx.x >> 6), nwaves = gridDim.x * 4u;
  for (uint32_t i = wave_id; i < count; i += nwaves) {
    const uint2 item = a.tile_list[i];
    TriRaster r;
    tri_prepare(a.big_list[item.x], a.width, a.height, r);
    const uint32_t ntx = (uint32_t)((r.px1 - r.px0 + kBigTile) / kBigTile);
    const int64_t ox = r.px0 + (int64_t)(item.y % ntx) * kBigTile, oy = r.py0 + (int64_t)(item.y / ntx) * kBigTile;
    raster_rect(a, r, ox, oy, min(ox + kBigTile - 1, r.px1), min(oy + kBigTile - 1, r.py1), lane);
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
  (void)hipMemsetAsync(a.clip_count, 0, kRasterHeaderBytes, s);  // clip / tile counters and the big list's segment counters
  hipLaunchKernelGGL(k_draw_rows, dim3(std::max(1u, std::min((a.mesh_instance_count + 255u) / 256u, max_grid))), dim3(256), 0, s, a);
  if (a.wide == 2u) {  // {id, corner} pairs
    hipLaunchKernelGGL(k_draw_setup<true>, dim3(max_grid), dim3(256), 0, s, a);
    hipLaunchKernelGGL((k_draw_clipped<false, true>), dim3(256), dim3(64), 0, s, a);
    hipLaunchKernelGGL((k_draw_clipped<true, true>), dim3(max_grid), dim3(64), 0, s, a);
  } else {
    hipLaunchKernelGGL(k_draw_setup<false>, dim3(max_grid), dim3(256), 0, s, a);
    hipLaunchKernelGGL((k_draw_clipped<false, false>), dim3(256), dim3(64), 0, s, a);
    hipLaunchKernelGGL((k_draw_clipped<true, false>), dim3(max_grid), dim3(64), 0, s, a);  // (returns at once unless the id queue overflowed)
  }
  hipLaunchKernelGGL(k_draw_big, dim3(max_grid), dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_draw_big_tiles, dim3(max_grid), dim3(256), 0, s, a);
  if (depth_out || vis_out)
    hipLaunchKernelGGL(k_resolve_visbuffer, dim3((uint32_t)std::min<uint64_t>((n + 255) / 256, max_grid)), dim3(256), 0, s, a.visdepth, n, depth_out, vis_out);
}

}  // namespace oxc
