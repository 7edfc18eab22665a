// oxcull_kernels.hip -- the gfx950 kernels of the meshlet visibility pipeline.
//
// Replaces the reference compute pipelines hiz / cull_meshes / cull_meshlets / cull_meshlets_hiz /
// cull_triangles (Oxylus/src/Render/Shaders/passes/*.slang).  The reference allocates output
// slots with global / LDS atomics (order is a race); here every stage is
//     test  -> 64-bit wave ballots (1 bit per candidate) + per-chunk survivor counts
//     emit  -> ordered expansion of the ballots into the output list
// so outputs are ascending, deterministic, and no contended atomic exists anywhere.
// The only atomics left are per-super-chunk count accumulations (64 adds per address) and
// the partial-word updates of the persistent visibility mask.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "oxcull_device.hpp"
#include "oxcull_kernels.hpp"
#include "oxcull_types.hpp"

#pragma clang fp contract(off)

namespace oxc {

// ------------------------------------------------------------------------------------------
// small wave/block helpers (64-lane waves)
// ------------------------------------------------------------------------------------------
OXC_DEV uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// inclusive prefix sum across the 64 lanes
OXC_DEV uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t t = __shfl_up(v, o, 64);
    if (lane >= o) v += t;
  }
  return v;
}
// sum over a 256-thread block; result valid in every thread.  s_red: 4 words of LDS.
OXC_DEV uint32_t block_sum_256(uint32_t v, uint32_t* s_red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}
// Exclusive base of chunk `c`: all supers before its super + the chunk counts inside it.
OXC_DEV uint32_t chunk_base_256(const uint32_t* __restrict__ supers, const uint32_t* __restrict__ chunk_counts, uint32_t c,
                                uint32_t* s_red) {
  uint32_t s = c / kChunksPerSuper;
  uint32_t acc = 0;
  for (uint32_t i = threadIdx.x; i < s; i += 256) acc += gptr(supers)[i * kSuperStride];
  uint32_t j = s * kChunksPerSuper + threadIdx.x;
  if (j < c) acc += gptr(chunk_counts)[j];  // at most 63 terms
  return block_sum_256(acc, s_red);
}

// ------------------------------------------------------------------------------------------
// k_prepare_instances: per mesh instance, derive the InstCache; optionally run cull_meshes'
// frustum + LOD select (passes/cull_meshes.slang:17-58).  Also (re)initialises the counter
// slot like the reference's scratch_buffer initial values (CullGeometry.cpp:97-100,125-127,
// 380-382) and zeroes the super-chunk accumulators.
// ------------------------------------------------------------------------------------------
OXC_DEV void prepare_body(const PrepareArgs& a, const uint32_t view) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t nthreads = gridDim.x * blockDim.x;
  // view 0: the cull camera; 1 + v: VSM clipmap v (use_hpb)
  float pv[16];
  if (view == 0) {
#pragma unroll
    for (int k = 0; k < 16; k++) pv[k] = a.cam.projection_view[k];
  } else {
#pragma unroll
    for (int k = 0; k < 16; k++) pv[k] = a.clipmaps[view - 1].projection_view_mat[k];
  }
  InstCache* const rows = view == 0 ? a.cache : a.view_cache + (size_t)(view - 1) * a.mesh_instance_count;
  const bool main_view = view == 0;
  if (main_view && tid == 0) {
    a.slot[SLOT_TRI_CMD + 0] = 0;
    a.slot[SLOT_TRI_CMD + 1] = 1;
    a.slot[SLOT_TRI_CMD + 2] = 1;
    a.slot[SLOT_DRAW_CMD + 0] = 0;
    a.slot[SLOT_DRAW_CMD + 1] = 1;
    a.slot[SLOT_DRAW_CMD + 2] = 0;
    a.slot[SLOT_DRAW_CMD + 3] = 0;
    a.slot[SLOT_DRAW_CMD + 4] = 0;
    if (a.init_vis) {
      a.vis[0] = a.seed_total;
      a.vis[1] = 0;
      a.vis[2] = 0;
      a.meshlets_cmd[0] = (a.seed_total + 63u) / 64u;
      a.meshlets_cmd[1] = 1;
      a.meshlets_cmd[2] = 1;
    }
  }
  if (main_view) {
    for (uint32_t i = tid; i < a.n_supers_meshlets; i += nthreads) a.supers_meshlets[i * kSuperStride] = 0;
    for (uint32_t i = tid; i < a.n_supers_tris; i += nthreads) a.supers_tris[i * kSuperStride] = 0;
    if (a.tickets && tid < kTicketCounters) a.tickets[tid * kSuperStride] = 0;
    if (a.slot_late) {  // (wave-uniform) the late call of this frame: its accumulators and counter slot, see PrepareArgs
      if (tid == 0) {
        a.slot_late[SLOT_TRI_CMD + 0] = 0;
        a.slot_late[SLOT_TRI_CMD + 1] = 1;
        a.slot_late[SLOT_TRI_CMD + 2] = 1;
        a.slot_late[SLOT_DRAW_CMD + 0] = 0;
        a.slot_late[SLOT_DRAW_CMD + 1] = 1;
        a.slot_late[SLOT_DRAW_CMD + 2] = 0;
        a.slot_late[SLOT_DRAW_CMD + 3] = 0;
        a.slot_late[SLOT_DRAW_CMD + 4] = 0;
      }
      for (uint32_t i = tid; i < a.n_supers_meshlets; i += nthreads) a.supers_meshlets_late[i * kSuperStride] = 0;
      for (uint32_t i = tid; i < a.n_supers_tris; i += nthreads) a.supers_tris_late[i * kSuperStride] = 0;
      if (tid < kTicketCounters) a.tickets_late[tid * kSuperStride] = 0;
    }
  }
  const bool do_cull_meshes = main_view && a.do_cull_meshes;

  // Eight lanes per mesh instance: lanes 0..5 each normalise one frustum plane (the sqrt +
  // 4 divides are the long pole), lane 6 writes mvp + world rows, lane 7 the normal matrix,
  // scale, LOD selection and the resolved LOD pointers.  Uniform code, lane-dependent data.
  // The row is put together in LDS and leaves as three 16-byte stores per lane: one 128-byte piece per instance and store instruction.  (Through
  // round 3 every field went out on its own: ~80 store instructions per wave, each touching eight rows -- the kernel was bound by that, not
  // by its arithmetic or its dependent loads: batched prepare of 16 views x 10 K instances 38.6 -> 27.2 us)
  __shared__ __attribute__((aligned(64))) InstCache s_rows[32];  // (256 threads per block: k_prepare_instances / k_prepare_batch)
  const int lane = threadIdx.x & 63;
  const uint32_t sub = tid & 7u;
  const uint32_t ngroups = nthreads >> 3;
  const uint32_t rounds = (a.mesh_instance_count + ngroups - 1) / ngroups;
  for (uint32_t round = 0; round < rounds; round++) {
    const uint32_t mi = round * ngroups + (tid >> 3);
    const bool valid = mi < a.mesh_instance_count;
    GpuMeshInstance inst = {0, 0, 0, 0, 0};
    if (valid) inst = a.mesh_instances[mi];
    GpuMesh mesh = {};
    float w[16], mvp[16];
    if (valid) {
      mesh = a.meshes[inst.mesh_index];
      const float4* wp = reinterpret_cast<const float4*>(a.transforms + (size_t)inst.transform_index * 16);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        float4 v = wp[k];
        w[k * 4 + 0] = v.x;
        w[k * 4 + 1] = v.y;
        w[k * 4 + 2] = v.z;
        w[k * 4 + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 16; k++) w[k] = (k % 5 == 0) ? 1.0f : 0.0f;
    }
    mul_mat4(pv, w, mvp);

    // plane `sub` (cull.slang:58-71): {r3+r0, r3-r0, r3+r1, r3-r1, r2, r3-r2}.  x - y == x + (-y)
    // exactly, so the sign is data; plane 4 is a select, not an add.
    const uint32_t sel = sub >> 1;  // row 0,0,1,1,2,2 (lanes 6,7: row 3, unused)
    const bool neg = (sub & 1u) != 0u;
    float pl[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
      float r0 = OXC_M(mvp, 0, c), r1 = OXC_M(mvp, 1, c), r2 = OXC_M(mvp, 2, c), r3 = OXC_M(mvp, 3, c);
      float rk = sel == 0 ? r0 : (sel == 1 ? r1 : r2);
      float sum = r3 + (neg ? -rk : rk);
      pl[c] = sub == 4 ? r2 : sum;
    }
    {
      float l = len3(pl[0], pl[1], pl[2]);
#pragma unroll
      for (int c = 0; c < 4; c++) pl[c] = pl[c] / l;
    }
    InstCache* out = &s_rows[threadIdx.x >> 3];
    if (valid && sub < 6) {
#pragma unroll
      for (int c = 0; c < 4; c++) out->planes2[sub >> 1][c][sub & 1u] = pl[c];
#pragma unroll
      for (int c = 0; c < 3; c++) out->signs2[sub >> 1][c][sub & 1u] = (asu(pl[c]) & 0x80000000u) ? -1.0f : 1.0f;
    }
    // Clipmap rows (view >= 1): does this view's frustum have the plane NORMALS of clipmap 0's, bit for bit?  (The clipmaps of one light
    // do: orthographic matrices that differ by a power-of-two scale and a translation -- the normalised normals are the same floats, only
    // the plane distances differ.)  k_cull_meshlets_hpb_test then computes a box's six plane distances once and compares them per view.
    // Plane `sub` of clipmap 0 is derived again here, with the same arithmetic, and compared; the flag takes the row's vis_offset slot
    // (the visibility offset is a property of the instance: the kernels read it from the camera's row).
    uint32_t same_normals = 0u;
    if (view >= 1) {
      bool same = true;
      if (view >= 2) {
        float mvp0[16], p0[3];
        mul_mat4(a.clipmaps[0].projection_view_mat, w, mvp0);
#pragma unroll
        for (int c = 0; c < 3; c++) {
          float r0 = OXC_M(mvp0, 0, c), r1 = OXC_M(mvp0, 1, c), r2 = OXC_M(mvp0, 2, c), r3 = OXC_M(mvp0, 3, c);
          float rk = sel == 0 ? r0 : (sel == 1 ? r1 : r2);
          float sum = r3 + (neg ? -rk : rk);
          p0[c] = sub == 4 ? r2 : sum;
        }
        const float l0 = len3(p0[0], p0[1], p0[2]);
#pragma unroll
        for (int c = 0; c < 3; c++) same = same && asu(p0[c] / l0) == asu(pl[c]);
      }
      const uint64_t diff = __ballot(sub < 6 && !same);
      same_normals = ((diff >> (lane & ~7)) & 0x3Full) == 0ull ? 1u : 0u;
    }

    // cull_meshes frustum test of the mesh AABB (cull_meshes.slang:34): lane k tests plane k
    bool outside = false;
    if (do_cull_meshes) {
      float hx = mesh.aabb_extent[0] * 0.5f, hy = mesh.aabb_extent[1] * 0.5f, hz = mesh.aabb_extent[2] * 0.5f;
      float qx = mesh.aabb_center[0] + asf(asu(hx) ^ (asu(pl[0]) & 0x80000000u));
      float qy = mesh.aabb_center[1] + asf(asu(hy) ^ (asu(pl[1]) & 0x80000000u));
      float qz = mesh.aabb_center[2] + asf(asu(hz) ^ (asu(pl[2]) & 0x80000000u));
      outside = sub < 6 && (dot3(qx, qy, qz, pl[0], pl[1], pl[2]) <= -pl[3]);
    }
    const uint64_t out_bits = __ballot(outside);
    const bool in_frustum = ((out_bits >> (lane & ~7)) & 0x3Full) == 0ull;

    if (valid && sub == 6) {
#pragma unroll
      for (int k = 0; k < 4; k++)
        *reinterpret_cast<float4*>(&out->mvp[k * 4]) = make_float4(mvp[k * 4], mvp[k * 4 + 1], mvp[k * 4 + 2], mvp[k * 4 + 3]);
#pragma unroll
      for (int c = 0; c < 3; c++) {
        out->world2[c][0] = OXC_M(w, 0, c);
        out->world2[c][1] = OXC_M(w, 1, c);
      }
      out->world_t2[0] = OXC_M(w, 0, 3);
      out->world_t2[1] = OXC_M(w, 1, 3);
#pragma unroll
      for (int c = 0; c < 4; c++) out->world_r2[c] = OXC_M(w, 2, c);
    }
    if (valid && sub == 7) {
      float nm[9];
      normal_matrix(w, nm);
#pragma unroll
      for (int k = 0; k < 9; k++) out->nm[k] = nm[k];
      float sx = len3(OXC_M(w, 0, 0), OXC_M(w, 0, 1), OXC_M(w, 0, 2));
      float sy = len3(OXC_M(w, 1, 0), OXC_M(w, 1, 1), OXC_M(w, 1, 2));
      float sz = len3(OXC_M(w, 2, 0), OXC_M(w, 2, 1), OXC_M(w, 2, 2));
      out->scale_max = fmaxf(sx, fmaxf(sy, sz));
      out->vis_offset = view >= 1 ? same_normals : inst.meshlet_instance_visibility_offset;
      out->transform_index = inst.transform_index;
      out->_pad0[0] = out->_pad0[1] = out->_pad0[2] = 0u;  // zero dwords: dummy index source for degenerate meshlets (tris_test_body)

      const GpuMeshLOD* lods = reinterpret_cast<const GpuMeshLOD*>(mesh.lods);
      uint32_t lod_index = inst.lod_index;
      if (do_cull_meshes) {
        uint32_t meshlet_count = 0;
        lod_index = 0;
        if ((a.cull_flags & OXC_CULL_TEST_FRUSTUM) && in_frustum) {
          if (a.cull_flags & OXC_CULL_SELECT_LOD) {  // cull_meshes.slang:35-57
            float cx = mesh.aabb_center[0], cy = mesh.aabb_center[1], cz = mesh.aabb_center[2];
            float ex = mesh.aabb_extent[0], ey = mesh.aabb_extent[1], ez = mesh.aabb_extent[2];
            float wc[3], we[3];
#pragma unroll
            for (int r = 0; r < 3; r++) {
              wc[r] = ((OXC_M(w, r, 0) * cx + OXC_M(w, r, 1) * cy) + OXC_M(w, r, 2) * cz) + OXC_M(w, r, 3);
              we[r] = fabsf(((OXC_M(w, r, 0) * ex + OXC_M(w, r, 1) * ey) + OXC_M(w, r, 2) * ez) + OXC_M(w, r, 3) * 0.0f);
            }
            float rough = fmaxf(we[0], fmaxf(we[1], we[2]));
            float dx = wc[0] - a.cam.position[0], dy = wc[1] - a.cam.position[1], dz = wc[2] - a.cam.position[2];
            float dist = fmaxf(len3(dx, dy, dz) - 0.5f * rough, 0.0f);
            float pixel_size_at_1m = 2.0f / fmaxf(a.cam.resolution[0], a.cam.resolution[1]);
            float size_at_1m = rough / dist;
            float px = size_at_1m / pixel_size_at_1m;
            for (uint32_t i = 1; i < mesh.lod_count; i++) {
              float err = px * lods[i].error;
              if (err < a.cam.acceptable_lod_error)
                lod_index = i;
              else
                break;
            }
          }
          meshlet_count = lods[lod_index].meshlet_count;
        }
        a.mesh_counts[mi] = meshlet_count;
        if (meshlet_count > 0) a.mesh_instances[mi].lod_index = lod_index;  // cull_meshes.slang:76
      }
      const GpuMeshLOD lod = lods[lod_index];
      out->meshlet_count = lod.meshlet_count;
      out->bounds = lod.meshlet_bounds;
      out->meshlets = lod.meshlets;
      out->micro = lod.local_triangle_indices;
      out->vidx = lod.indirect_vertex_indices;
      out->positions = mesh.vertex_positions;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same-wave LDS hand-off (the eight lanes of an instance sit in one wave)
    if (valid) {
      const uint4* src = reinterpret_cast<const uint4*>(out);
      uint4* dst = reinterpret_cast<uint4*>(rows + mi);
      static_assert(sizeof(InstCache) == 24 * 16, "three 16-byte pieces per lane");
#pragma unroll
      for (uint32_t k = 0; k < 3u; k++) dst[sub + 8u * k] = src[sub + 8u * k];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the LDS rows are rewritten by the next round
  }
}

// ------------------------------------------------------------------------------------------
// cull_meshes expansion (passes/cull_meshes.slang:60-84), deterministic: exclusive scan of the
// per-instance meshlet counts, then one wave per instance writes its MeshletInstance records.
// ------------------------------------------------------------------------------------------
OXC_DEV void scan_body(const ScanArgs& a) {
  const uint32_t* __restrict__ counts = a.counts;
  uint32_t* __restrict__ offsets = a.offsets;
  const uint32_t n = a.n;
  uint32_t* __restrict__ vis = a.vis;
  uint32_t* __restrict__ meshlets_cmd = a.meshlets_cmd;
  __shared__ uint32_t s_wave[16];
  __shared__ uint32_t s_carry;
  __shared__ uint32_t s_part[16 * 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr uint32_t kTiles = 16;
  if (n <= kTiles * 1024u) {
    // Up to 16K mesh instances (the usual case): all loads in flight at once, one wave scan per 1024-element tile,
    // then ONE block-level scan of the 256 (tile, wave) totals -- three barriers in all, and no load waits behind a
    // barrier (the tile loop below costs ~1.5 us of latency per tile: 15 us for 10K instances, this path ~4 us).
    uint32_t v[kTiles], incl[kTiles];
#pragma unroll
    for (uint32_t k = 0; k < kTiles; k++) {
      const uint32_t i = k * 1024u + threadIdx.x;
      v[k] = i < n ? counts[i] : 0u;
    }
#pragma unroll
    for (uint32_t k = 0; k < kTiles; k++) {
      incl[k] = wave_incl_scan(v[k], lane);
      if (lane == 63) s_part[k * 16 + wave] = incl[k];
    }
    __syncthreads();
    uint32_t p = 0, pin = 0;
    if (threadIdx.x < 256) {  // totals in element order: index = tile * 16 + wave
      p = s_part[threadIdx.x];
      pin = wave_incl_scan(p, lane);
      if (lane == 63) s_wave[wave] = pin;
    }
    __syncthreads();
    if (threadIdx.x < 256) {
      uint32_t woff = 0;
      for (int k = 0; k < wave; k++) woff += s_wave[k];
      s_part[threadIdx.x] = woff + pin - p;  // exclusive
      if (threadIdx.x == 255) s_carry = woff + pin;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < kTiles; k++) {
      const uint32_t i = k * 1024u + threadIdx.x;
      if (i < n) offsets[i] = s_part[k * 16 + wave] + incl[k] - v[k];
    }
  } else {
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
      uint32_t i = base + threadIdx.x;
      uint32_t v = i < n ? counts[i] : 0u;
      uint32_t incl = wave_incl_scan(v, lane);
      if (lane == 63) s_wave[wave] = incl;
      __syncthreads();
      uint32_t wave_off = 0;
      for (int k = 0; k < wave; k++) wave_off += s_wave[k];
      uint32_t carry = s_carry;
      if (i < n) offsets[i] = carry + wave_off + incl - v;
      __syncthreads();
      if (threadIdx.x == 1023) s_carry = carry + wave_off + incl;
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) {
    // (a total beyond what the caller's buffers hold is cut off there: expand_body writes no record past `cap`)
    uint32_t total = min(s_carry, a.cap);
    vis[0] = total;                       // visibility[0].total_visible_meshlet_instances
    meshlets_cmd[0] = (total + 63u) / 64u;  // atomic_max of ceil(new_total/64), cull_meshes.slang:68-70
  }
}

OXC_DEV void expand_body(const ExpandArgs& a) {
  const uint32_t* __restrict__ counts = a.counts;
  const uint32_t* __restrict__ offsets = a.offsets;
  const uint32_t n = a.n;
  GpuMeshletInstance* __restrict__ out = a.out;
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t mi = wave; mi < n; mi += nwaves) {
    uint32_t cnt = counts[mi], off = offsets[mi];
    if (off >= a.cap) continue;
    cnt = min(cnt, a.cap - off);
    for (uint32_t k = lane; k < cnt; k += 64) {
      // (`nt` stores here were measured on configs[4], 466 MB of records per step: this kernel 86 -> 120 us, the kernels after it 20 us
      //  faster in total -- a net loss; the records that nobody reads are better not written at all: implicit_meshlet_instances)
      GpuMeshletInstance r;
      r.mesh_instance_index = mi;
      r.meshlet_index = k;
      out[off + k] = r;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Meshlet stage, test kernels.  passes/cull_meshlets.slang:23-73 (plain) and
// passes/cull_meshlets_hiz.slang:19-88 (HiZ variants).
// One lane per meshlet instance, 64 consecutive instances per group, G groups per wave step.
// A wave whose meshlets span several mesh instances runs one round per distinct instance.
// ------------------------------------------------------------------------------------------
// Set/clear bits of the persistent visibility mask for the lanes in `active`.
// Lanes whose (idx - lane) agree form a "run": lane l owns global bit (d + l), so a run is a
// 64-bit window at bit offset d and touches at most three mask words.  Whole words are stored,
// partial words use atomic and/or (disjoint bits from other waves).  cull_meshlets_hiz.slang:81-87.
constexpr uint32_t kMaskNone = 0xFFFFFFFFu;  // "this lane has no bit in the caller's mask buffer"
OXC_DEV void update_visibility_mask(uint32_t* __restrict__ mask, uint32_t idx, bool visible, bool active, int lane) {
  uint64_t rem = __ballot(active);
  const int32_t d_l = (int32_t)(idx - (uint32_t)lane);
  // The usual case -- 64 consecutive meshlets of one instance (or of instances whose offsets follow each other), window on a word
  // boundary: the wave owns two whole words; lanes 0 and 1 store them.  Same stores the general loop below would issue, ~10
  // instructions instead of ~45 per group (the read-modify-write was 11 of the late kernel's 122 us).
  if (rem == ~0ull) {
    const int32_t d0 = __builtin_amdgcn_readfirstlane(d_l);
    if ((d0 & 31) == 0 && __ballot(d_l != d0) == 0ull) {
      const uint64_t vis = __ballot(visible);
      if (lane < 2) mask[(d0 >> 5) + lane] = lane == 0 ? (uint32_t)vis : (uint32_t)(vis >> 32);
      return;
    }
  }
  while (rem) {
    int leader = __ffsll((unsigned long long)rem) - 1;
    int32_t d = __builtin_amdgcn_readlane(d_l, leader);
    bool in_run = active && d_l == d;
    uint64_t run = __ballot(in_run) & rem;
    uint64_t vis = __ballot(in_run && visible) & rem;
    int32_t w0 = d >> 5;  // arithmetic shift: floor(d / 32)
    uint32_t s = (uint32_t)d & 31u;
    if (lane < 3) {
      uint64_t lo_r = run << s, lo_v = vis << s;
      uint32_t hi_r = s ? (uint32_t)(run >> (64 - s)) : 0u;
      uint32_t hi_v = s ? (uint32_t)(vis >> (64 - s)) : 0u;
      uint32_t clr = lane == 0 ? (uint32_t)lo_r : (lane == 1 ? (uint32_t)(lo_r >> 32) : hi_r);
      uint32_t set = lane == 0 ? (uint32_t)lo_v : (lane == 1 ? (uint32_t)(lo_v >> 32) : hi_v);
      int32_t w = w0 + lane;
      if (clr != 0u) {
        if (clr == 0xFFFFFFFFu) {
          mask[w] = set;
        } else {
          uint32_t zero = clr & ~set;
          if (zero) atomicAnd(&mask[w], ~zero);
          if (set) atomicOr(&mask[w], set);
        }
      }
    }
    rem &= ~run;
  }
}

// ------------------------------------------------------------------------------------------
// Meshlet stage, plain variant (passes/cull_meshlets.slang:23-73): the configs[1] kernel.
// Organised around the two scarce resources measured on gfx950: VALU issue slots and SGPRs.
//  * Instance-constant operands reach the SGPRs by SCALAR loads (s_load_dwordx16 from the InstCache row
//    through the constant address space): zero VALU instructions, where a vector load + v_readlane
//    unpack costs one VALU slot per dword.
//  * Two phases per instance round, so that the 42 frustum operands and the 26 cone operands are never
//    live together (the one-phase kernel spills ~65 SGPRs to VGPR lanes and pays a v_readlane per use):
//    phase 1 decodes the bounds and runs the packed frustum test for all G groups, phase 2 loads the cone
//    operands and runs the cone test for the groups where a frustum survivor still needs it.
//  * Per-lane state lives in VGPRs (plentiful: 512 per SIMD lane, the kernel needs < 128), not in lane masks.
// ------------------------------------------------------------------------------------------
typedef const uint32_t __attribute__((address_space(4))) * kconst32p;
OXC_DEV kconst32p const_row(const InstCache* cache, uint32_t mi) { return (kconst32p)(reinterpret_cast<uint64_t>(cache + mi)); }

// MeshletInstance records and MeshletBounds are read exactly once per call: `nt` loads (measured on configs[1]:
// test kernel 28.0 -> 27.4 us per 4M meshlets, whole job +2.5 %).
#define OXC_LOAD_MLI load_stream_u2
#define OXC_LOAD_BND load_stream_u4
template <int G, bool UNORD = false, uint32_t kWaves = kPlainBlockWaves>
OXC_DEV void meshlets_plain_body(const MeshletTestArgs& a) {
  set_half_denorm_flush();
  constexpr uint32_t kStep = kWaves * G * 64;  // meshlets per block iteration
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t N = a.n_host ? a.n_host : min(gptr(a.vis)[0], a.n_cap);
  const uint32_t nwords = (N + 63u) / 64u;
  const uint32_t nchunks = (N + kStep - 1) / kStep;
  const uint64_t mlis = reinterpret_cast<uint64_t>(a.meshlet_instances);
  const uint32_t last_index = N ? N - 1u : 0u;
  const float camx = a.cam_pos[0], camy = a.cam_pos[1], camz = a.cam_pos[2];

  for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    // ---- stage A: all MeshletInstance loads of this wave.  Unconditional at a clamped index: a
    // `cond ? load : x` is lowered to an exec-masked block with s_waitcnt vmcnt(0) inside, which
    // serialises the G loads (one HBM round trip each)
    const uint32_t group0 = (chunk * kWaves + wave) * G;
    uint2 rec[G];
    uint32_t st[G];  // bit 0: still to be decided, bit 1: visible
#pragma unroll
    for (int j = 0; j < G; j++) rec[j] = OXC_LOAD_MLI(mlis, min((group0 + j) * 64 + lane, last_index));
#pragma unroll
    for (int j = 0; j < G; j++) st[j] = ((group0 + j) * 64 + lane < N) ? 1u : 0u;

    for (;;) {
      // leader = first undecided lane of the first group that has one (wave-uniform)
      uint32_t mi_u = 0;
      bool found = false;
#pragma unroll
      for (int j = 0; j < G; j++) {
        const uint64_t p = __builtin_amdgcn_ballot_w64((st[j] & 1u) != 0u);
        if (!found && p) {
          mi_u = readlane_u(rec[j].x, __ffsll((unsigned long long)p) - 1);
          found = true;
        }
      }
      if (!found) break;
      const kconst32p row = const_row(a.cache, mi_u);
      const uint64_t bounds = (uint64_t)row[kRowBounds] | ((uint64_t)row[kRowBounds + 1] << 32);
      uint4 bnd[G];
      bool mine[G];
#pragma unroll
      for (int j = 0; j < G; j++) {
        mine[j] = (st[j] & 1u) != 0u && rec[j].x == mi_u;
        bnd[j] = OXC_LOAD_BND(bounds, mine[j] ? rec[j].y : 0u);  // other lanes read element 0 (always valid)
      }
      // ---- phase 1: bounds decode + frustum
      float cx[G], cy[G], cz[G], ex[G], ey[G], ez[G];
      uint32_t need[G];
      uint64_t any_need = 0;
      {
        float pl[24], sg[18];
#pragma unroll
        for (int k = 0; k < 24; k++) pl[k] = asf(row[kRowPlanes + k]);
#pragma unroll
        for (int k = 0; k < 18; k++) sg[k] = asf(row[kRowSigns + k]);
#pragma unroll
        for (int j = 0; j < G; j++) {
          const uint4 b = bnd[j];
          cx[j] = dequantize_half(b.x & 0xFFFFu), cy[j] = dequantize_half(b.x >> 16), cz[j] = dequantize_half(b.y & 0xFFFFu);
          ex[j] = dequantize_half(b.z & 0xFFFFu), ey[j] = dequantize_half(b.z >> 16), ez[j] = dequantize_half(b.w & 0xFFFFu);
          const bool vis = mine[j] & test_frustum_planes(pl, sg, cx[j], cy[j], cz[j], ex[j], ey[j], ez[j]);
          // cutoff >= 1.0 <=> s8 == 127: cone test skipped (cull_meshlets.slang:52)
          const bool nc = vis & (((int32_t)b.w >> 24) != 127);
          need[j] = nc ? 1u : 0u;
          any_need |= __builtin_amdgcn_ballot_w64(nc);
          st[j] = mine[j] ? (vis ? 2u : 0u) : st[j];
          // keeps the scheduler from interleaving all four groups' frustum math: no SGPR spills left (6 before), -2 %
          if (j == 1) __builtin_amdgcn_sched_barrier(0);
        }
      }
      // ---- phase 2: normal cone, only when some frustum survivor of this instance needs it
      if (any_need) {
        ConeU cu;
#pragma unroll
        for (int k = 0; k < 9; k++) cu.nm[k] = asf(row[kRowNm + k]);
#pragma unroll
        for (int k = 0; k < 6; k++) cu.w2[k >> 1][k & 1] = asf(row[kRowWorld2 + k]);
#pragma unroll
        for (int k = 0; k < 2; k++) cu.wt2[k] = asf(row[kRowWorldT2 + k]);
#pragma unroll
        for (int k = 0; k < 4; k++) cu.wr2[k] = asf(row[kRowWorldR2 + k]);
        cu.scale_max = asf(row[kRowScale]);
#pragma unroll
        for (int j = 0; j < G; j++) {
          if (__builtin_amdgcn_ballot_w64(need[j] != 0u) == 0) continue;  // wave-uniform
          uint4 b = bnd[j];
          // Decode the centre / extent again instead of keeping 24 decoded floats of phase 1 alive across the phases (the
          // opaque asm stops the optimiser from recognising the repeat and keeping them anyway): ~9 more VALU per group for
          // ~24 fewer VGPRs at the pressure peak.
          asm volatile("" : "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w));
          const float qx = dequantize_half(b.x & 0xFFFFu), qy = dequantize_half(b.x >> 16), qz = dequantize_half(b.y & 0xFFFFu);
          const float rx = dequantize_half(b.z & 0xFFFFu), ry = dequantize_half(b.z >> 16), rz = dequantize_half(b.w & 0xFFFFu);
          const f2 axy = s8_over_127_x2((int32_t)(b.y << 8) >> 24, (int32_t)b.y >> 24);
          const f2 azc = s8_over_127_x2((int32_t)(b.w << 8) >> 24, (int32_t)b.w >> 24);
          const int tier1 = cone_visible_fast(cu, camx, camy, camz, qx, qy, qz, rx, ry, rz, axy.x, axy.y, azc.x, azc.y);
          bool cone_ok = tier1 == 1;
          if (__builtin_amdgcn_ballot_w64(need[j] != 0u && tier1 == 2)) {  // some lane sits within the margin: the canonical IEEE path decides
            const bool exact = cone_visible(cu, camx, camy, camz, qx, qy, qz, rx, ry, rz, axy.x, axy.y, azc.x, azc.y);
            cone_ok = tier1 == 2 ? exact : cone_ok;
          }
          st[j] = (need[j] != 0u && !cone_ok) ? 0u : st[j];
        }
      }
    }
    if constexpr (UNORD) {
      // ---- unordered_output: the block appends its survivors itself.  cull_meshlets.slang:55-70 with the workgroup's LDS slot
      // counter replaced by the wave ballots and ONE returning atomic_add per block iteration (1024 meshlets; the reference: one per
      // 64) on cull_triangles_cmd.x -- a single address retires ~88 atomics per microsecond on this part, so the aggregation is what
      // keeps a 1 M-meshlet call from queueing on it.  Order inside a block's run is ascending; the runs land in arrival order.
      __shared__ uint32_t s_cnt[2][kWaves];
      __shared__ uint32_t s_base[2];
      const uint32_t par = ((chunk - blockIdx.x) / gridDim.x) & 1u;  // (two sets: a wave may be one iteration ahead of the slowest reader)
      uint64_t bits[G];
      uint32_t cnt = 0;
#pragma unroll
      for (int j = 0; j < G; j++) {
        bits[j] = __builtin_amdgcn_ballot_w64((st[j] & 2u) != 0u);
        cnt += (uint32_t)__popcll((unsigned long long)bits[j]);
      }
      if (lane == 0) s_cnt[par][wave] = cnt;
      __syncthreads();
      // every wave scans the block's wave counts itself (lane w holds wave w's): its own offset and the block total come out of one scan
      const uint32_t incl = wave_incl_scan((uint32_t)lane < kWaves ? s_cnt[par][(uint32_t)lane < kWaves ? lane : 0] : 0u, lane);
      const uint32_t total = readlane_u(incl, (int)kWaves - 1);
      if (threadIdx.x == 0) s_base[par] = total ? __hip_atomic_fetch_add(gptr(a.count_a), total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
      __syncthreads();
      uint32_t at = s_base[par] + (wave ? readlane_u(incl, wave - 1) : 0u);
#pragma unroll
      for (int j = 0; j < G; j++) {
        if ((bits[j] >> lane) & 1ull)
          gptr(a.out)[at + __builtin_amdgcn_mbcnt_hi((uint32_t)(bits[j] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bits[j], 0u))] = (group0 + j) * 64 + (uint32_t)lane;
        at += (uint32_t)__popcll((unsigned long long)bits[j]);
      }
    } else {
    // ---- ballots + per-WAVE survivor count (+ per-super accumulation), published by lane 0 without a
    // block barrier: a __syncthreads() here re-couples the block's waves every iteration (measured ~25 %)
    uint32_t cnt = 0;
#pragma unroll
    for (int j = 0; j < G; j++) {
      if (group0 + j >= nwords) continue;  // wave-uniform
      const uint64_t bits = __builtin_amdgcn_ballot_w64((st[j] & 2u) != 0u);
      if (lane == 0) gptr(a.bits)[group0 + j] = bits;
      cnt += (uint32_t)__popcll((unsigned long long)bits);
    }
    if (lane == 0 && group0 < nwords) {
      const uint32_t wchunk = chunk * kWaves + wave;  // counts are per 64*G meshlets
      gptr(a.chunk_counts)[wchunk] = cnt;
      if (cnt) __hip_atomic_fetch_add(gptr(a.supers) + (wchunk / kChunksPerSuper) * kSuperStride, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    }
  }
}

// ------------------------------------------------------------------------------------------
// Meshlet stage, HiZ variants (passes/cull_meshlets_hiz.slang:19-88), same organisation as
// meshlets_plain_body plus a third phase: occlusion (mvp operands, project_aabb + HiZ fetch) for the
// groups that still have a visible lane.  OCCL = TestOcclusion, LATE = LatePass.
// ------------------------------------------------------------------------------------------
// ticket t of counter x (of K) -> wave step.  Four consecutive tickets of a counter are four consecutive steps (usually one mesh
// instance: its InstCache row, mask words and pyramid texels stay in the L2 of the XCD the counter's blocks run on; measured ~1 %).
constexpr uint32_t kTicketRun = 4;
OXC_DEV uint32_t OXC_TICKET_STEP(uint32_t t, uint32_t K, uint32_t x) { return ((t / kTicketRun) * K + x) * kTicketRun + t % kTicketRun; }
// COUNT (measurement aid, oxc_debug_count_occlusion_candidates): the same kernel also adds the number of candidates that reach
// test_occlusion -- SURVEY 8d's f, each costs four pyramid taps = 16 B -- to MeshletTestArgs::dbg_occlusion; a separate instantiation,
// so that the kernels that are timed do not carry the pointer.
template <bool OCCL, bool LATE, int G, int SHARE = 0, bool COUNT = false>
OXC_DEV void meshlets_hiz_body(const MeshletTestArgs& a) {
  // SHARE (only with OCCL, G == 4): 1 = early call that also runs the cone test for the meshlets that were not visible last frame and
  // publishes the "passed frustum and cone" ballots and each step's mask run, 2 = late call that takes both from the early call of the
  // same frame (MeshletTestArgs::share, include/oxcull.h: share_pass_tests)
  static_assert(SHARE == 0 || (OCCL && G == 4 && (SHARE == 1) == !LATE), "sharing: early call writes, late call reads");
  set_half_denorm_flush();
  constexpr int kWaves = 16 / G;
  constexpr bool OCCL_OR_LATE = OCCL || LATE;  // HAS_FLAG(flags, TestOcclusion|LatePass) is "any of"
  __shared__ uint32_t s_level_off[13];
  __shared__ uint32_t s_lds_off[13];
  __shared__ float s_hiz_top[kHizLdsTexels];
  __shared__ uint4 s_strip[OCCL_OR_LATE ? kWaves : 1][G * 64];  // frustum survivors (then cone survivors) of one round, per wave
  __shared__ uint32_t s_flags[OCCL_OR_LATE ? kWaves : 1][G * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t N = a.n_host ? a.n_host : min(gptr(a.vis)[0], a.n_cap);
  const uint32_t nwords = (N + 63u) / 64u;
  const uint32_t nchunks = (N + kMeshletChunk - 1) / kMeshletChunk;
  if (threadIdx.x < 13) {
    s_level_off[threadIdx.x] = a.hiz_level_off[threadIdx.x];
    s_lds_off[threadIdx.x] = a.hiz_lds_off[threadIdx.x];
  }
  // stage the top of the pyramid (levels >= hiz_lds_first) once per block
  for (uint32_t k = a.hiz_lds_first; k < a.hiz_levels; k++) {
    const uint32_t n = mip_dim(a.hiz_w, k) * mip_dim(a.hiz_h, k);
    const float* src = a.hiz_data + a.hiz_level_off[k];
    float* dst = s_hiz_top + a.hiz_lds_off[k];
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  HizView hiz;
  hiz.data = a.hiz_data;
  hiz.width = a.hiz_w;
  hiz.height = a.hiz_h;
  hiz.levels = a.hiz_levels;
  hiz.lds = s_hiz_top;
  hiz.lds_off = s_lds_off;
  hiz.lds_first = a.hiz_lds_first;
  hiz.inv_width = exact_reciprocal_or_zero(a.hiz_w);
  hiz.inv_height = exact_reciprocal_or_zero(a.hiz_h);
  const uint64_t mlis = reinterpret_cast<uint64_t>(a.meshlet_instances);
  const uint32_t last_index = N ? N - 1u : 0u;
  const float camx = a.cam_pos[0], camy = a.cam_pos[1], camz = a.cam_pos[2];

  // Work distribution.  A wave step (64 * G meshlets) costs anything between a few hundred cycles (early pass, nothing of it was visible
  // last frame) and several thousand (a visible instance: cone + occlusion), so a fixed stride leaves the CUs half empty while
  // the unlucky waves finish (SQ_WAVE_CYCLES / SQ_BUSY_CU_CYCLES: 13.8 of 16 resident waves on average, the tail on a nearly empty machine).  With
  // a.tickets the waves draw their steps from kTicketCounters counters instead -- counter x = blockIdx % K hands out
  // the steps congruent to x mod K (K = min(kTicketCounters, grid)); the next ticket is requested while the current step is worked on.  Results are
  // stored by step index, so the output is the same whatever wave does the step.  Same-address atomics retire at ~13 ns each on
  // this part (one counter: 520 us per launch), hence many counters: 8 -> 116 us, 32 -> 82, 256 -> 79 (fixed stride: 87), 512 -> 83;
  // two steps per ticket or drawing the ticket later in the step are worse (config 3, early pass; the late pass 113 -> 101 us).
  const uint32_t nsteps = nchunks * kWaves;
  const uint32_t K = min(kTicketCounters, gridDim.x), kx = blockIdx.x % K;  // every counter in use has at least one block drawing from it
  uint32_t* const ticket = a.tickets ? a.tickets + kx * kSuperStride : nullptr;
  auto draw_ticket = [&]() -> uint32_t {
    uint32_t t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return t;
  };
  uint32_t step = blockIdx.x * kWaves + wave;
  if (ticket) step = OXC_TICKET_STEP(readlane_u(draw_ticket(), 0), K, kx);
  while (step < nsteps) {
    const uint32_t group0 = step * G;
    uint2 rec[G];
    uint32_t st[G];        // bit 0: still to be decided, bit 1: visible, bit 2: was_visible
    uint32_t mask_idx[G];  // bit index into the persistent visibility mask
    // SHARE: bit 3 of st = "passed the frustum and the cone test" (SHARE == 1: found by this call and published at the end of the step;
    // SHARE == 2: read from what the early call of the frame published)
    bool quick = false;  // SHARE == 2: nothing of this step passed the camera tests and its mask bits are one run: no record is needed
    uint32_t fbit[G];
#pragma unroll
    for (int j = 0; j < G; j++) fbit[j] = 0u;
    if constexpr (SHARE == 2) {
      uint64_t any = 0;
#pragma unroll
      for (int j = 0; j < G; j++) {
        const uint64_t w = group0 + j < nwords ? gptr(a.camera_test_bits)[group0 + j] : 0ull;
        any |= w;
        fbit[j] = ((w >> lane) & 1ull) != 0ull ? 8u : 0u;
      }
      const uint2 info = load_global_u2(reinterpret_cast<uint64_t>(a.step_info), step);
      quick = any == 0ull && info.y != 0u;
      if (quick) {
#pragma unroll
        for (int j = 0; j < G; j++) {
          st[j] = 0u;
          mask_idx[j] = info.x + 64u * (uint32_t)j + (uint32_t)lane;
          rec[j] = make_uint2(0u, 0u);
        }
      }
    }
    if (!quick) {
#pragma unroll
      for (int j = 0; j < G; j++) rec[j] = OXC_LOAD_MLI(mlis, min((group0 + j) * 64 + lane, last_index));
    }
    const uint32_t next_ticket = ticket ? draw_ticket() : 0u;  // in flight behind the record loads; read at the end of the step
    if (!quick) {
#pragma unroll
      for (int j = 0; j < G; j++) {
        st[j] = (((group0 + j) * 64 + lane < N) ? 1u : 0u) | fbit[j];
        mask_idx[j] = 0;
      }
    }
    for (;;) {
      uint32_t mi_u = 0;
      bool found = false;
#pragma unroll
      for (int j = 0; j < G; j++) {
        const uint64_t p = __builtin_amdgcn_ballot_w64((st[j] & 1u) != 0u);
        if (!found && p) {
          mi_u = readlane_u(rec[j].x, __ffsll((unsigned long long)p) - 1);
          found = true;
        }
      }
      if (!found) break;
      const kconst32p row = const_row(a.cache, mi_u);
      const uint64_t bounds = (uint64_t)row[kRowBounds] | ((uint64_t)row[kRowBounds + 1] << 32);
      const uint32_t vis_offset = row[kRowVisOffset];
      uint4 bnd[G];
      bool mine[G];
      uint32_t mword[G];
#pragma unroll
      for (int j = 0; j < G; j++) {
        mine[j] = (st[j] & 1u) != 0u && rec[j].x == mi_u;
        // (SHARE == 2: only the meshlets that passed frustum and cone in the early call are looked at again)
        const bool fetch = SHARE == 2 ? (mine[j] && (st[j] & 8u) != 0u) : mine[j];
        bnd[j] = OXC_LOAD_BND(bounds, fetch ? rec[j].y : 0u);  // other lanes read element 0 (always valid)
        mword[j] = 0xFFFFFFFFu;
        if (OCCL) {  // cull_meshlets_hiz.slang:45-51 (unconditional load: lanes of other instances re-read this instance's first word)
          // (a mask index beyond the caller's buffer -- inconsistent visibility offsets -- reads as "not visible" and is
          // never written: kMaskNone)
          uint32_t mi_bit = vis_offset + (mine[j] ? rec[j].y : 0u);
          const bool in_mask = mi_bit < a.mask_bits;
          mi_bit = in_mask ? mi_bit : 0u;
          mask_idx[j] = mine[j] ? (in_mask ? mi_bit : kMaskNone) : mask_idx[j];
          mword[j] = load_global_u32(reinterpret_cast<uint64_t>(a.mask), mi_bit >> 5) >> (mi_bit & 31u);
          mword[j] = in_mask ? mword[j] : 0u;
        }
      }
      // ---- phase 1: bounds decode + frustum
      float cx[G], cy[G], cz[G], ex[G], ey[G], ez[G];
      uint32_t need[G];
      uint64_t any_need = 0;
      {
        float pl[24], sg[18];
#pragma unroll
        for (int k = 0; k < 24; k++) pl[k] = asf(row[kRowPlanes + k]);
#pragma unroll
        for (int k = 0; k < 18; k++) sg[k] = asf(row[kRowSigns + k]);
#pragma unroll
        for (int j = 0; j < G; j++) {
          const uint4 b = bnd[j];
          const bool was_visible = (mword[j] & 1u) != 0u;
          if constexpr (SHARE == 2) {  // the early call of this frame ran the same frustum and cone tests on the same operands
            const bool vis = mine[j] & ((st[j] & 8u) != 0u);
            cx[j] = cy[j] = cz[j] = ex[j] = ey[j] = ez[j] = 0.0f;
            need[j] = 0u;
            st[j] = mine[j] ? ((vis ? 2u : 0u) | (was_visible ? 4u : 0u)) : st[j];
            continue;
          }
          if (!LATE && SHARE == 0 && __builtin_amdgcn_ballot_w64(mine[j] && was_visible) == 0) {
            // early pass, and no meshlet of this group was visible last frame (cull_meshlets_hiz.slang:45-51: they all return
            // before any test): skip the decode + frustum code for the whole group.  Visibility is coherent per instance, so
            // this is most groups of a typical frame.
            cx[j] = cy[j] = cz[j] = ex[j] = ey[j] = ez[j] = 0.0f;
            need[j] = 0u;
            st[j] = mine[j] ? 0u : st[j];
            continue;
          }
          cx[j] = dequantize_half(b.x & 0xFFFFu), cy[j] = dequantize_half(b.x >> 16), cz[j] = dequantize_half(b.y & 0xFFFFu);
          ex[j] = dequantize_half(b.z & 0xFFFFu), ey[j] = dequantize_half(b.z >> 16), ez[j] = dequantize_half(b.w & 0xFFFFu);
          const bool inside = mine[j] & test_frustum_planes(pl, sg, cx[j], cy[j], cz[j], ex[j], ey[j], ez[j]);
          // (SHARE == 1: every meshlet inside the frustum goes on to the cone test, whose result the late call reuses; bit 1 means
          //  "visible" only after the scatter below)
          const bool vis = inside & ((LATE || SHARE == 1) ? true : was_visible);
          const bool nc = vis & (((int32_t)b.w >> 24) != 127);  // cutoff >= 1.0 <=> s8 == 127: cone test skipped
          need[j] = nc ? 1u : 0u;
          any_need |= __builtin_amdgcn_ballot_w64(nc);
          st[j] = mine[j] ? ((vis ? 2u : 0u) | (was_visible ? 4u : 0u)) : st[j];
        }
      }
      // ---- phases 2 + 3: normal cone, then occlusion against the pyramid (cull_meshlets_hiz.slang:52-66).  Both run on DENSE lanes:
      // the frustum survivors of the round's G groups (10-40 % of the lanes) are compacted through a per-wave LDS strip, the cone test
      // runs over ceil(survivors / 64) batches and compacts ITS survivors in place (a survivor's record moves to a position at or below
      // its own, behind what the batch has already read), and the occlusion test -- 8 projected corners, 24 divisions: by far the longest
      // piece of straight-line code -- runs over ceil(cone survivors / 64) batches.  Round 2 ran the cone test once per group on sparse
      // lanes (4 passes per step) and compacted only in front of the occlusion test; compacting once and running cone + occlusion back to
      // back on the frustum survivors was measured too and is slower (more occlusion passes: 78 -> 83 / 102 -> 108 us).  All lanes of a
      // round share the instance, hence the cone and mvp operands.
      if constexpr (OCCL_OR_LATE) {
        uint64_t vb[G];
        uint32_t base[G + 1];
        base[0] = 0;
#pragma unroll
        for (int j = 0; j < G; j++) {
          vb[j] = __builtin_amdgcn_ballot_w64(mine[j] && (st[j] & 2u) != 0u);
          base[j + 1] = base[j] + (uint32_t)__popcll((unsigned long long)vb[j]);
        }
        const uint32_t total = base[G];
        if (total) {
          uint4* strip = s_strip[wave];
          uint32_t* flg = s_flags[wave];  // per frustum survivor (strip position): still visible?
          uint32_t slot[G];
#pragma unroll
          for (int j = 0; j < G; j++) {
            slot[j] = base[j] + __builtin_amdgcn_mbcnt_hi((uint32_t)(vb[j] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vb[j], 0u));
            if ((vb[j] >> lane) & 1ull) {
              if constexpr (SHARE == 2) {  // these passed frustum and cone in the early call: straight to the occlusion batches
                strip[slot[j]] = make_uint4(bnd[j].x, bnd[j].y, bnd[j].z, (bnd[j].w & 0x00FFFFFFu) | (slot[j] << 24));
                flg[slot[j]] = 1u;
              } else {
                strip[slot[j]] = bnd[j];
                if constexpr (SHARE == 1) flg[slot[j]] = st[j] & 4u;  // was_visible travels with the record
              }
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same-wave LDS hand-off: in order, no barrier needed
          uint32_t nb = SHARE == 2 ? total : 0u;  // occlusion candidates: they occupy strip[0 .. nb)
          if constexpr (SHARE != 2) {
          ConeU cu;
#pragma unroll
          for (int k = 0; k < 9; k++) cu.nm[k] = asf(row[kRowNm + k]);
#pragma unroll
          for (int k = 0; k < 6; k++) cu.w2[k >> 1][k & 1] = asf(row[kRowWorld2 + k]);
#pragma unroll
          for (int k = 0; k < 2; k++) cu.wt2[k] = asf(row[kRowWorldT2 + k]);
#pragma unroll
          for (int k = 0; k < 4; k++) cu.wr2[k] = asf(row[kRowWorldR2 + k]);
          cu.scale_max = asf(row[kRowScale]);
          for (uint32_t t0 = 0; t0 < total; t0 += 64) {
            const uint32_t t = t0 + (uint32_t)lane;
            const bool act = t < total;
            const uint4 b = strip[act ? t : total - 1u];
            bool ok = act;
            const bool nc = act && ((int32_t)b.w >> 24) != 127;  // cutoff >= 1.0 <=> s8 == 127: cone test skipped (cull_meshlets.slang:52)
            if (__builtin_amdgcn_ballot_w64(nc)) {
              const float qx = dequantize_half(b.x & 0xFFFFu), qy = dequantize_half(b.x >> 16), qz = dequantize_half(b.y & 0xFFFFu);
              const float rx = dequantize_half(b.z & 0xFFFFu), ry = dequantize_half(b.z >> 16), rz = dequantize_half(b.w & 0xFFFFu);
              const f2 axy = s8_over_127_x2((int32_t)(b.y << 8) >> 24, (int32_t)b.y >> 24);
              const f2 azc = s8_over_127_x2((int32_t)(b.w << 8) >> 24, (int32_t)b.w >> 24);
              const int tier1 = cone_visible_fast(cu, camx, camy, camz, qx, qy, qz, rx, ry, rz, axy.x, axy.y, azc.x, azc.y);
              bool cone_ok = tier1 == 1;
              if (__builtin_amdgcn_ballot_w64(nc && tier1 == 2)) {  // some lane sits within the margin: the canonical IEEE path decides
                const bool exact = cone_visible(cu, camx, camy, camz, qx, qy, qz, rx, ry, rz, axy.x, axy.y, azc.x, azc.y);
                cone_ok = tier1 == 2 ? exact : cone_ok;
              }
              ok = nc ? cone_ok : ok;
            }
            // SHARE == 1: the occlusion test is for the meshlets that were visible last frame; bit 0 of the flag = passed frustum and
            // cone (published for the late call), bit 1 = visible so far
            const bool okv = SHARE == 1 ? (ok && (flg[act ? t : 0u] & 4u) != 0u) : ok;
            const uint64_t okb = __builtin_amdgcn_ballot_w64(okv);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(okb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)okb, 0u));
            if (act) flg[t] = SHARE == 1 ? ((ok ? 1u : 0u) | (okv ? 2u : 0u)) : (ok ? 1u : 0u);
            // (nb + rank <= t, and every lane of this batch has read its record: the LDS queue of a wave is in order)
            if (okv) strip[nb + rank] = make_uint4(b.x, b.y, b.z, (b.w & 0x00FFFFFFu) | (t << 24));  // the cutoff byte now carries the survivor's strip position
            nb += (uint32_t)__popcll((unsigned long long)okb);
          }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if constexpr (COUNT) {
            if (nb && lane == 0) __hip_atomic_fetch_add(gptr(a.dbg_occlusion) + (step & 255u) * kSuperStride, nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if (nb) {
            float mvp[16];
#pragma unroll
            for (int k = 0; k < 16; k++) mvp[k] = asf(row[kRowMvp + k]);
            for (uint32_t u0 = 0; u0 < nb; u0 += 64) {
              const uint32_t u = u0 + (uint32_t)lane;
              const bool act = u < nb;
              const uint4 b = strip[act ? u : nb - 1u];
              const float qx = dequantize_half(b.x & 0xFFFFu), qy = dequantize_half(b.x >> 16), qz = dequantize_half(b.y & 0xFFFFu);
              const float rx = dequantize_half(b.z & 0xFFFFu), ry = dequantize_half(b.z >> 16), rz = dequantize_half(b.w & 0xFFFFu);
              const bool occluded = aabb_occluded(mvp, a.near_clip, qx, qy, qz, rx, ry, rz, hiz, s_level_off, act);
              if (act && occluded) flg[b.w >> 24] = SHARE == 1 ? 1u : 0u;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          }
#pragma unroll
          for (int j = 0; j < G; j++) {
            if ((vb[j] >> lane) & 1ull) {
              const uint32_t f = flg[slot[j]];
              if constexpr (SHARE == 1)
                st[j] = (st[j] & ~2u) | (f & 2u) | ((f & 1u) << 3);
              else
                st[j] = f ? st[j] : (st[j] & ~2u);
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // strip and flags are rewritten by the next round
        }
      } else if (any_need) {  // no occlusion phase: the cone test where it stands, one pass per group that needs it
        ConeU cu;
#pragma unroll
        for (int k = 0; k < 9; k++) cu.nm[k] = asf(row[kRowNm + k]);
#pragma unroll
        for (int k = 0; k < 6; k++) cu.w2[k >> 1][k & 1] = asf(row[kRowWorld2 + k]);
#pragma unroll
        for (int k = 0; k < 2; k++) cu.wt2[k] = asf(row[kRowWorldT2 + k]);
#pragma unroll
        for (int k = 0; k < 4; k++) cu.wr2[k] = asf(row[kRowWorldR2 + k]);
        cu.scale_max = asf(row[kRowScale]);
#pragma unroll
        for (int j = 0; j < G; j++) {
          if (__builtin_amdgcn_ballot_w64(need[j] != 0u) == 0) continue;  // wave-uniform
          uint4 b = bnd[j];
          asm volatile("" : "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w));  // decode again instead of keeping phase 1's floats alive (see meshlets_plain_body)
          const float qx = dequantize_half(b.x & 0xFFFFu), qy = dequantize_half(b.x >> 16), qz = dequantize_half(b.y & 0xFFFFu);
          const float rx = dequantize_half(b.z & 0xFFFFu), ry = dequantize_half(b.z >> 16), rz = dequantize_half(b.w & 0xFFFFu);
          const f2 axy = s8_over_127_x2((int32_t)(b.y << 8) >> 24, (int32_t)b.y >> 24);
          const f2 azc = s8_over_127_x2((int32_t)(b.w << 8) >> 24, (int32_t)b.w >> 24);
          const int tier1 = cone_visible_fast(cu, camx, camy, camz, qx, qy, qz, rx, ry, rz, axy.x, axy.y, azc.x, azc.y);
          bool cone_ok = tier1 == 1;
          if (__builtin_amdgcn_ballot_w64(need[j] != 0u && tier1 == 2)) {
            const bool exact = cone_visible(cu, camx, camy, camz, qx, qy, qz, rx, ry, rz, axy.x, axy.y, azc.x, azc.y);
            cone_ok = tier1 == 2 ? exact : cone_ok;
          }
          st[j] = (need[j] != 0u && !cone_ok) ? (st[j] & ~2u) : st[j];
        }
      }
    }
    // ---- mask update, ballots, per-wave survivor count
    // The usual step is 64 * G consecutive meshlets of one instance: its mask bits are ONE run of 64 * G bits at bit offset d0 -- 2 G
    // whole words (+ one partial word at either end when d0 is not a multiple of 32).  Lane k builds word k from the G visibility
    // ballots and the run is written in one go, instead of once per 64-meshlet group through update_visibility_mask (~40 VALU
    // instructions each, 13 % of a late step's instructions).  Same stores / atomics as the per-group path would issue, merged.
    bool step_run = false;
    uint32_t run_first = 0;
    if constexpr (OCCL && G == 4) {
      if ((group0 + G) * 64u <= N) {  // (wave-uniform)
        const uint32_t d0 = readfirst_u(mask_idx[0]);
        run_first = d0;
        uint64_t bad = 0;
#pragma unroll
        for (int j = 0; j < G; j++) bad |= __builtin_amdgcn_ballot_w64(mask_idx[j] != d0 + 64u * (uint32_t)j + (uint32_t)lane);  // (kMaskNone lanes differ)
        if (bad == 0ull && d0 <= 0xFFFFFFFFu - 64u * G) {
          step_run = true;
          // the G ballots go through the wave's LDS flag row (free between rounds): lane k then picks dword k and dword k - 1 of the run
          uint32_t* row8 = s_flags[wave];
#pragma unroll
          for (int j = 0; j < G; j++) {
            const uint64_t vb = __builtin_amdgcn_ballot_w64((st[j] & 2u) != 0u);
            if (lane == 0) {
              row8[2 * j] = (uint32_t)vb;
              row8[2 * j + 1] = (uint32_t)(vb >> 32);
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same-wave LDS hand-off
          const uint32_t cur = lane < 2 * G ? row8[lane] : 0u;
          const uint32_t prev = (lane >= 1 && lane <= 2 * G) ? row8[lane - 1] : 0u;
          const uint32_t sh = d0 & 31u, w0 = d0 >> 5;
          if (sh == 0u) {
            if (lane < 2 * G) a.mask[w0 + (uint32_t)lane] = cur;
          } else {
            const uint32_t word = __builtin_amdgcn_alignbit(cur, prev, 32u - sh);  // (cur << sh) | (prev >> (32 - sh))
            if (lane >= 1 && lane < 2 * G) {
              a.mask[w0 + (uint32_t)lane] = word;
            } else if (lane == 0 || lane == 2 * G) {  // the run's first / last word: only its own bits
              const uint32_t own = lane == 0 ? (0xFFFFFFFFu << sh) : (0xFFFFFFFFu >> (32u - sh));
              const uint32_t zero = own & ~word;
              if (zero) atomicAnd(&a.mask[w0 + (uint32_t)lane], ~zero);
              if (word) atomicOr(&a.mask[w0 + (uint32_t)lane], word);
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the flag row is rewritten by the next step
        }
      }
    }
    if constexpr (SHARE == 1) {
#pragma unroll
      for (int j = 0; j < G; j++) {
        const uint64_t w = __builtin_amdgcn_ballot_w64((st[j] & 8u) != 0u);
        if (lane == 0 && group0 + j < nwords) gptr(a.camera_test_bits)[group0 + j] = w;
      }
      if (lane == 0) {
        uint32_t* info = reinterpret_cast<uint32_t*>(a.step_info + step);
        gptr(info)[0] = run_first;
        gptr(info)[1] = step_run ? 1u : 0u;
      }
    }
    uint32_t cnt = 0;
#pragma unroll
    for (int j = 0; j < G; j++) {
      if (group0 + j >= nwords) continue;  // wave-uniform
      const bool visible = (st[j] & 2u) != 0u;
      // Every mask read of this wave step precedes its writes.  With TestOcclusion off the
      // reference's and/or hit word 0 with an empty bit (no-op), so nothing to do.
      if (OCCL && !step_run) update_visibility_mask(a.mask, mask_idx[j], visible, (group0 + j) * 64 + lane < N && mask_idx[j] != kMaskNone, lane);
      const bool emit = visible && (!LATE || (st[j] & 4u) == 0u);
      const uint64_t bits = __builtin_amdgcn_ballot_w64(emit);
      if (lane == 0) gptr(a.bits)[group0 + j] = bits;
      cnt += (uint32_t)__popcll((unsigned long long)bits);
    }
    // (Round 4 also built the reference's literal slot allocation here -- cull_meshlets_hiz.slang:67-78: an atomic_add on the early / late
    // counter and one on cull_triangles_cmd.x, aggregated to one pair per wave step through the ballots -- as unordered_output = 2:
    // 150 / 154 us per launch against 79 + 11 / 66 + 11 for this kernel + the ordered emit; every step with a survivor queues on the same
    // two addresses.  Removed in round 5: git log -S count_b.)
    if (lane == 0 && group0 < nwords) {
      gptr(a.chunk_counts)[step] = cnt;
      if (cnt) __hip_atomic_fetch_add(gptr(a.supers) + (step / kChunksPerSuper) * kSuperStride, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    step = ticket ? OXC_TICKET_STEP(readlane_u(next_ticket, 0), K, kx) : step + gridDim.x * kWaves;
  }
}

// ------------------------------------------------------------------------------------------
// VSM multi-view meshlet test (passes/cull_meshlets_hpb.slang:25-99): directional cone +
// camera frustum, then "visible if ANY dirty clipmap view passes frustum + page-pyramid test".
// The reference's `break` on the first visible view has no side effect, so the result is the OR
// over dirty views; views a whole wave no longer needs are skipped.  Output: ballots, expanded
// by k_cull_meshlets_emit<false,false> like the plain meshlet stage.
// ------------------------------------------------------------------------------------------
constexpr int kHpbWaves = 6;  // waves per SIMD (80 VGPRs, 7 spilled dwords): 338-341 us per 10 M meshlets x 10 views against 347-353 at 5 and 348 at 4
__global__ __launch_bounds__(256, kHpbWaves) void k_cull_meshlets_hpb_test(HpbTestArgs a) {
  // Round 2: the view loop is the OUTER loop of a wave step.  Round 1 walked one 64-meshlet group at a time and, inside it, one
  // clipmap view at a time, so every (group, view) paid the scalar-load round trips of the view's plane / matrix row on its own --
  // 0.89 ms per 10 M meshlets, latency-bound.  Here a wave holds G groups (all loads batched, as in the plain kernel) and a view's
  // row is fetched once per instance round for all of them.
  // The page test of a view (project_aabb: 8 corners, 24 divisions, then the pyramid fetch) runs on the lanes that passed the view's
  // frustum -- a minority of most groups -- so those lanes are first compacted across the wave's G groups through an LDS strip, as
  // the HiZ kernel does for its occlusion phase, and waves draw their steps from the ticket counters (a step costs between one frustum
  // test and ten projections).
  set_half_denorm_flush();
  __shared__ uint32_t s_level_off[13];
  constexpr int G = (int)kGroupsPerWave;
  constexpr uint32_t kWaves = 4;
  __shared__ uint4 s_strip[kWaves][G * 64];   // the candidates of a round (camera frustum + cone survivors), dense
  __shared__ uint32_t s_strip2[kWaves][G * 64];  // of those, the strip positions of the ones inside the current view's frustum: the page test's lanes
  __shared__ uint32_t s_seen[kWaves][G * 64];  // per candidate: some view's pages want it
  __shared__ uint64_t s_in[kWaves][16][G];     // per view and 64-candidate batch: the candidates inside the view's frustum
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t N = min(a.vis[0], a.n_cap);
  const uint32_t nwords = (N + 63u) / 64u;
  const uint32_t nchunks = (N + kMeshletChunk - 1) / kMeshletChunk;
  if (threadIdx.x < 13) s_level_off[threadIdx.x] = a.hpb_level_off[threadIdx.x];
  __syncthreads();
  const uint64_t dirty_mask = __builtin_amdgcn_ballot_w64((uint32_t)lane < a.clipmap_count && a.dirty[(uint32_t)lane < a.clipmap_count ? lane : 0] != 0u);
  HpbView hpb;
  hpb.data = a.hpb_data;
  hpb.width = a.hpb_w;
  hpb.height = a.hpb_h;
  hpb.layers = a.hpb_layers;
  hpb.levels = a.hpb_levels;
  const uint64_t mlis = reinterpret_cast<uint64_t>(a.meshlet_instances);
  const uint32_t last_index = N ? N - 1u : 0u;

  const uint32_t nsteps = nchunks * kWaves;
  const uint32_t K = min(kTicketCounters, gridDim.x), kx = blockIdx.x % K;
  uint32_t* const ticket = a.tickets + kx * kSuperStride;
  auto draw_ticket = [&]() -> uint32_t {
    uint32_t t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return t;
  };
  uint4* const strip = s_strip[wave];
  uint32_t* const strip2 = s_strip2[wave];
  uint32_t* const seen = s_seen[wave];
  uint32_t step = OXC_TICKET_STEP(readlane_u(draw_ticket(), 0), K, kx);
  while (step < nsteps) {
    const uint32_t group0 = step * G;
    uint2 rec[G];
    uint32_t st[G];  // bit 0: still to be decided, bit 1: visible
#pragma unroll
    for (int j = 0; j < G; j++) rec[j] = load_stream_u2(mlis, min((group0 + j) * 64 + lane, last_index));
    const uint32_t next_ticket = draw_ticket();
#pragma unroll
    for (int j = 0; j < G; j++) st[j] = ((group0 + j) * 64 + lane < N) ? 1u : 0u;
    for (;;) {
      uint32_t mi_u = 0;
      bool found = false;
#pragma unroll
      for (int j = 0; j < G; j++) {
        const uint64_t p = __builtin_amdgcn_ballot_w64((st[j] & 1u) != 0u);
        if (!found && p) {
          mi_u = readlane_u(rec[j].x, __ffsll((unsigned long long)p) - 1);
          found = true;
        }
      }
      if (!found) break;
      const kconst32p row = const_row(a.cache, mi_u);
      const uint64_t bounds = (uint64_t)row[kRowBounds] | ((uint64_t)row[kRowBounds + 1] << 32);
      uint4 bnd[G];
      bool mine[G], cand[G], vis[G];
#pragma unroll
      for (int j = 0; j < G; j++) {
        mine[j] = (st[j] & 1u) != 0u && rec[j].x == mi_u;
        bnd[j] = load_stream_u4(bounds, mine[j] ? rec[j].y : 0u);  // other lanes read element 0 (always valid)
      }
      // (centre / extent are decoded from the 16-byte record at every use: 6 conversions against 24 VGPRs held across the view loop)
#define OXC_HPB_DECODE(b)                                                                                                             \
  const float cxj = dequantize_half((b).x & 0xFFFFu), cyj = dequantize_half((b).x >> 16), czj = dequantize_half((b).y & 0xFFFFu); \
  const float exj = dequantize_half((b).z & 0xFFFFu), eyj = dequantize_half((b).z >> 16), ezj = dequantize_half((b).w & 0xFFFFu)
      uint64_t any_cand = 0;
      {  // camera frustum (cull_meshlets_hpb.slang:49-52)
        float pl[24], sg[18];
#pragma unroll
        for (int k = 0; k < 24; k++) pl[k] = asf(row[kRowPlanes + k]);
#pragma unroll
        for (int k = 0; k < 18; k++) sg[k] = asf(row[kRowSigns + k]);
#pragma unroll
        for (int j = 0; j < G; j++) {
          OXC_HPB_DECODE(bnd[j]);
          cand[j] = mine[j] & test_frustum_planes(pl, sg, cxj, cyj, czj, exj, eyj, ezj);
          vis[j] = false;
          any_cand |= __builtin_amdgcn_ballot_w64(cand[j]);
        }
      }
      if (any_cand) {  // cull_meshlets_hpb.slang:53-54: directional cone
        float nm[9];
#pragma unroll
        for (int k = 0; k < 9; k++) nm[k] = asf(row[kRowNm + k]);
        any_cand = 0;
#pragma unroll
        for (int j = 0; j < G; j++) {
          if (__builtin_amdgcn_ballot_w64(cand[j]) == 0) continue;  // wave-uniform
          const uint4 b = bnd[j];
          const f2 axy = s8_over_127_x2((int32_t)(b.y << 8) >> 24, (int32_t)b.y >> 24);
          const f2 azc = s8_over_127_x2((int32_t)(b.w << 8) >> 24, (int32_t)b.w >> 24);
          cand[j] = cand[j] && cone_visible_directional(nm, a.light_dir[0], a.light_dir[1], a.light_dir[2], axy.x, axy.y, azc.x, azc.y);
          any_cand |= __builtin_amdgcn_ballot_w64(cand[j]);
        }
      }
      // "visible if ANY dirty clipmap view passes frustum + page test" (:59-79); the reference's break on the first hit has no side effect.
      // The candidates of the round (camera frustum + cone survivors: a minority of the lanes) are compacted once through the LDS
      // strip; every view then runs its frustum over ceil(candidates / 64) DENSE batches (round 2 ran it over the G sparse groups),
      // compacts the strip positions of the lanes inside it and runs the page test (8 projected corners + the pyramid fetches) over
      // those.  A candidate some view has accepted drops out.
      if (any_cand) {
        uint64_t cb[G];
        uint32_t base[G + 1], slot[G];
        base[0] = 0;
#pragma unroll
        for (int j = 0; j < G; j++) {
          cb[j] = __builtin_amdgcn_ballot_w64(cand[j]);
          base[j + 1] = base[j] + (uint32_t)__popcll((unsigned long long)cb[j]);
        }
        const uint32_t total = base[G];
#pragma unroll
        for (int j = 0; j < G; j++) {
          slot[j] = base[j] + __builtin_amdgcn_mbcnt_hi((uint32_t)(cb[j] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cb[j], 0u));
          if (cand[j]) {
            strip[slot[j]] = bnd[j];
            seen[slot[j]] = 0u;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same-wave LDS hand-off: in order, no barrier needed
        uint32_t open_count = total;  // candidates no view has accepted yet (wave-uniform)
        // ---- phase A (round 4): the frustum of EVERY dirty view over the candidates, two batches at a time.  A box's six plane distances
        // are computed once, against clipmap 0's normals; a view whose row says "same normals" (prepare_body) only compares them with its
        // own plane offsets (7 instructions instead of ~90 per batch and view), any other view runs the whole test.  One ballot per
        // (view, batch) goes to LDS; the page test below picks its lanes from there.
        {
          const kconst32p rrow = const_row(a.view_cache, mi_u);
          float rpl[24], rsg[18];
#pragma unroll
          for (int k = 0; k < 24; k++) rpl[k] = asf(rrow[kRowPlanes + k]);
#pragma unroll
          for (int k = 0; k < 18; k++) rsg[k] = asf(rrow[kRowSigns + k]);
          // Lane v holds view v's six plane offsets and its "same normals" flag: one vector load round trip per instance round, the
          // view loop below then reads them with v_readlane instead of paying a scalar-load round trip per (view, chunk).
          float lw[6];
          uint64_t same_mask;
          {
            const uint32_t vl = (uint32_t)lane < a.clipmap_count ? (uint32_t)lane : 0u;
            const uint32_t* lrow = reinterpret_cast<const uint32_t*>(a.view_cache + (size_t)vl * a.mesh_instance_count + mi_u);
#pragma unroll
            for (int p = 0; p < 3; p++) {
              lw[2 * p] = -asf(lrow[kRowPlanes + p * 8 + 6]);
              lw[2 * p + 1] = -asf(lrow[kRowPlanes + p * 8 + 7]);
            }
            same_mask = __builtin_amdgcn_ballot_w64((uint32_t)lane < a.clipmap_count && lrow[kRowVisOffset] != 0u);
          }
          for (uint32_t c0 = 0; c0 < total; c0 += 128) {
            f2 dd[2][3];
            uint4 bb[2];
            bool actb[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
              const uint32_t t = c0 + 64u * (uint32_t)h + (uint32_t)lane;
              actb[h] = t < total;
              bb[h] = strip[actb[h] ? t : total - 1u];
              OXC_HPB_DECODE(bb[h]);
              frustum_plane_dots(rpl, rsg, cxj, cyj, czj, exj, eyj, ezj, dd[h]);
            }
            for (uint32_t v = 0; v < a.clipmap_count; v++) {
              if (((dirty_mask >> v) & 1ull) == 0ull) continue;  // uniform
              uint64_t ib[2];
              if ((same_mask >> v) & 1ull) {  // (wave-uniform) same plane normals as clipmap 0: only the offsets are this view's
                float nw[6];
#pragma unroll
                for (int k = 0; k < 6; k++) nw[k] = asf(__builtin_amdgcn_readlane((int)asu(lw[k]), (int)v));
#pragma unroll
                for (int h = 0; h < 2; h++) {
                  bool in = actb[h];
#pragma unroll
                  for (int p = 0; p < 3; p++) in = in & !(dd[h][p].x <= nw[2 * p]) & !(dd[h][p].y <= nw[2 * p + 1]);
                  ib[h] = __builtin_amdgcn_ballot_w64(in);
                }
              } else {
                // Any other view: its own frustum, one plane pair at a time (a rolled loop: the 14 scalars of a pair instead of the 42 of
                // the row next to clipmap 0's 42 -- unrolled, this branch alone cost the kernel 57 SGPR spills and 10 % of its time).
                const kconst32p vrow = const_row(a.view_cache + (size_t)v * a.mesh_instance_count, mi_u);
                bool inh[2] = {actb[0], actb[1]};
#pragma nounroll
                for (int p = 0; p < 3; p++) {
                  float n8[8], sg6[6];
#pragma unroll
                  for (int k = 0; k < 8; k++) n8[k] = asf(vrow[kRowPlanes + p * 8 + k]);
#pragma unroll
                  for (int k = 0; k < 6; k++) sg6[k] = asf(vrow[kRowSigns + p * 6 + k]);
#pragma unroll
                  for (int h = 0; h < 2; h++) {
                    OXC_HPB_DECODE(bb[h]);
                    inh[h] = inh[h] & frustum_pair_inside(n8, sg6, cxj, cyj, czj, exj, eyj, ezj);
                  }
                }
                ib[0] = __builtin_amdgcn_ballot_w64(inh[0]);
                ib[1] = __builtin_amdgcn_ballot_w64(inh[1]);
              }
              if (lane < 2) s_in[wave][v][(c0 >> 6) + (uint32_t)lane] = lane == 0 ? ib[0] : ib[1];
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same-wave LDS hand-off
        }
        for (uint32_t v = 0; v < a.clipmap_count && open_count; v++) {
          if (((dirty_mask >> v) & 1ull) == 0ull) continue;  // uniform
          const kconst32p vrow = const_row(a.view_cache + (size_t)v * a.mesh_instance_count, mi_u);
          uint32_t n_in = 0;  // lanes inside this view's frustum and not yet accepted by another view: strip2[0 .. n_in)
          for (uint32_t t0 = 0; t0 < total; t0 += 64) {
            const uint32_t t = t0 + (uint32_t)lane;
            const uint64_t inb = s_in[wave][v][t0 >> 6];
            const bool in = ((inb >> lane) & 1ull) != 0ull && seen[t < total ? t : 0u] == 0u;  // (a set bit implies t < total)
            const uint64_t ib = __builtin_amdgcn_ballot_w64(in);
            if (in) strip2[n_in + __builtin_amdgcn_mbcnt_hi((uint32_t)(ib >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ib, 0u))] = t;
            n_in += (uint32_t)__popcll((unsigned long long)ib);
          }
          if (n_in == 0) continue;
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          float vmvp[16];
#pragma unroll
          for (int k = 0; k < 16; k++) vmvp[k] = asf(vrow[kRowMvp + k]);
          const oxc_virtual_clipmap* cm = a.clipmaps + v;
          const float z_near = cm->z_near;
          const int32_t pox = cm->page_offset[0], poy = cm->page_offset[1];
          for (uint32_t t0 = 0; t0 < n_in; t0 += 64) {
            const uint32_t t = t0 + (uint32_t)lane;
            const bool act = t < n_in;
            const uint32_t at = strip2[act ? t : n_in - 1u];
            const uint4 b = strip[at];
            OXC_HPB_DECODE(b);
            float sa[6];
            bool pass = true;  // projection crosses the near plane: visible (cull_meshlets_hpb.slang:70-76)
            if (project_aabb<true, false>(vmvp, z_near, cxj, cyj, czj, exj, eyj, ezj, sa)) pass = test_vsm_page(sa, hpb, s_level_off, v, pox, poy);
            const bool hit = act && pass;
            if (hit) seen[at] = 1u;
            open_count -= (uint32_t)__popcll((unsigned long long)__builtin_amdgcn_ballot_w64(hit));
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the next view reads the flags and rewrites strip2
        }
#pragma unroll
        for (int j = 0; j < G; j++) {
          if (cand[j]) vis[j] = seen[slot[j]] != 0u;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // strip and flags are rewritten by the next round
      }
#pragma unroll
      for (int j = 0; j < G; j++) st[j] = mine[j] ? ((cand[j] && vis[j]) ? 2u : 0u) : st[j];
#undef OXC_HPB_DECODE
    }
    uint32_t cnt = 0;
#pragma unroll
    for (int j = 0; j < G; j++) {
      if (group0 + j >= nwords) continue;  // wave-uniform
      const uint64_t bits = __builtin_amdgcn_ballot_w64((st[j] & 2u) != 0u);
      if (lane == 0) gptr(a.bits)[group0 + j] = bits;
      cnt += (uint32_t)__popcll((unsigned long long)bits);
    }
    if (lane == 0 && group0 < nwords) {  // one count per wave step (64 * G meshlets), as the other meshlet test kernels publish them
      gptr(a.chunk_counts)[step] = cnt;
      if (cnt) __hip_atomic_fetch_add(gptr(a.supers) + (step / kChunksPerSuper) * kSuperStride, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    step = OXC_TICKET_STEP(readlane_u(next_ticket, 0), K, kx);
  }
}

// ------------------------------------------------------------------------------------------
// Meshlet stage, emit kernel: ordered expansion of the ballots into
// visible_meshlet_instances_indices (+ the indirect-dispatch / visibility counters).
// One block iteration = one span of 4096 candidates = 64 ballot words.
// ------------------------------------------------------------------------------------------
template <bool HIZ, bool LATE>
OXC_DEV void meshlets_emit_body(const MeshletEmitArgs& a) {
  // (One wave per span instead of one block -- the form k_mv_emit uses -- was measured here too: 10.9 -> 12.4 us per launch at 10 M
  // meshlets, every wave then sums all the super-chunk counts in front of its span on its own.)
  __shared__ uint32_t s_red[4];
  __shared__ uint32_t s_off[64];
  __shared__ uint64_t s_bits[64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t N = a.n_host ? a.n_host : min(gptr(a.vis)[0], a.n_cap);
  const uint32_t nwords = (N + 63u) / 64u;
  const uint32_t nspans = (N + kMeshletSpan - 1) / kMeshletSpan;
  const uint32_t out_first = (HIZ && LATE) ? gptr(a.vis)[1] : 0u;  // late list follows the early one (:73)
  const uint32_t counts_per_span = kMeshletSpan / a.count_meshlets;
  for (uint32_t span = blockIdx.x; span < nspans; span += gridDim.x) {
    const uint32_t base = chunk_base_256(a.supers, a.chunk_counts, span * counts_per_span, s_red);
    if (wave == 0) {
      uint32_t w = span * 64 + lane;
      uint64_t bits = gptr(a.bits)[min(w, nwords ? nwords - 1u : 0u)];
      if (w >= nwords) bits = 0ull;
      uint32_t c = (uint32_t)__popcll((unsigned long long)bits);
      uint32_t incl = wave_incl_scan(c, lane);
      s_off[lane] = incl - c;
      s_bits[lane] = bits;
      if (lane == 63 && span == nspans - 1) {
        uint32_t total = base + incl;
        gptr(a.tri_cmd)[0] = total;  // cull_triangles_cmd.x (one WG per visible meshlet in the reference)
        if (HIZ) gptr(a.vis)[LATE ? 2 : 1] = total;
      }
    }
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < 16; k++) {
      const int w = k * 4 + wave;
      const uint64_t bits = s_bits[w];
      if ((bits >> lane) & 1ull) {
        uint32_t rank = (uint32_t)__popcll((unsigned long long)(bits & ((1ull << lane) - 1ull)));
        gptr(a.out)[out_first + base + s_off[w] + rank] = (span * 64 + w) * 64 + lane;
      }
    }
    __syncthreads();
  }
}


// ------------------------------------------------------------------------------------------
// Expansion of a wave's consecutive slots (64-bit pass masks + meshlet-instance ids, both in LDS) into the packed index list
// (visbuffer.slang:13-14, cull_triangles.slang:82-88).  The slots' indices are contiguous in the output, so they are staged in
// rank order in a per-wave LDS run of kEmitRun dwords, laid out so that LDS index = global dword index (mod 4), and flushed as
// 16-byte stores per lane -- 1 KiB per wave instruction, whole 128-byte lines -- with at most three single dwords at either end.
// Round 3 stored every slot on its own (64 lanes x 4 bytes, 1.5 instructions per slot, lines shared between instructions): as `nt`
// stores those took twice as long, and as default-policy stores they left ~420 MB of dirty index lines per frame in L2 / the Infinity
// Cache, whose write-back the NEXT kernels paid (the pyramid build 69 -> 48 us and the triangle tests 90 / 164 -> 67 / 151 us with
// the lines gone, configs[2] frame).  Whole-line `nt` stores get both.  Same bytes at the same addresses as the per-slot form.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kEmitRun = 1024;  // dwords a wave stages between flushes (>= 384 + 3: one WIDE slot)
// PAIR (wide_triangle_index = 2, SURVEY A.7's form for shards beyond 2^23 ids): an index is the pair {u32 meshlet_instance_index, u32 corner} -- 8 bytes,
// six dwords per triangle, no id limit below 2^32.  g0 and the run count in DWORDS either way (the callers double the index offset); the offset is 64-bit
// in this form: 2^32 - 1 indices are 8.6e9 dwords.
template <int H, uint32_t kCornerBits, bool PAIR = false>
OXC_DEV void expand_slots_wide(const uint64_t* masks, const uint32_t* ids, int first_slot, int nslots, typename std::conditional<PAIR, uint64_t, uint32_t>::type g0,
                               uint32_t* __restrict__ out, uint32_t* run, int lane) {
  typedef uint32_t u4v __attribute__((ext_vector_type(4)));
  constexpr uint32_t kCornerMask = (1u << kCornerBits) - 1u;
  constexpr uint32_t kDw = PAIR ? 6u : 3u;  // dwords per emitted triangle
  static_assert(kEmitRun >= 128u * kDw + 3u || (!PAIR && H == 1), "a run holds at least one slot");
  uint32_t pad = (uint32_t)((reinterpret_cast<uint64_t>(out + g0) >> 2) & 3u);  // the run's first dword inside its 16-byte granule
  uint32_t filled = 0;
  auto flush = [&]() {
    // run[pad, pad + filled) -> out[g0, g0 + filled)
    const uint32_t head = min((4u - pad) & 3u, filled);
    if ((uint32_t)lane < head) out[g0 + (uint32_t)lane] = run[pad + (uint32_t)lane];
    const uint32_t nch = (filled - head) >> 2;
    const uint32_t* src = run + pad + head;  // 16-byte aligned in LDS (pad + head == 0 or 4 whenever nch > 0)
    uint32_t* dst = out + g0 + head;
    for (uint32_t c = (uint32_t)lane; c < nch; c += 64u) {
      const u4v v = *reinterpret_cast<const u4v*>(src + 4u * c);
      __builtin_nontemporal_store(v, reinterpret_cast<u4v*>(dst + 4u * c));
    }
    const uint32_t done = head + 4u * nch, tail = filled - done;
    if ((uint32_t)lane < tail) out[g0 + done + (uint32_t)lane] = run[pad + done + (uint32_t)lane];
    g0 += filled;
    pad = (pad + filled) & 3u;
    filled = 0;
  };
#pragma unroll 2
  for (int k = 0; k < nslots; k++) {
    const int s = first_slot + k;
    uint32_t cnt = 0;
    uint64_t m[H];
#pragma unroll
    for (int h = 0; h < H; h++) {
      m[h] = masks[s * H + h];
      cnt += (uint32_t)__popcll((unsigned long long)m[h]);
    }
    const uint32_t n3 = cnt * kDw;
    if (n3 == 0u) continue;                      // (wave-uniform)
    if (filled + n3 > kEmitRun) flush();         // (wave-uniform; same wave, in-order LDS: the flush has read the run before it is rewritten)
    uint32_t* at = run + pad + filled;
    uint32_t before = 0;
#pragma unroll
    for (int h = 0; h < H; h++) {
      if ((m[h] >> lane) & 1ull) {
        const uint32_t rank = before + (uint32_t)__popcll((unsigned long long)(m[h] & ((1ull << lane) - 1ull)));
        const uint32_t t3 = ((uint32_t)lane + 64u * (uint32_t)h) * 3u;
        if (PAIR) {
          const uint32_t id = ids[s];
#pragma unroll
          for (uint32_t c = 0; c < 3u; c++) {
            at[rank * 6u + 2u * c] = id;
            at[rank * 6u + 2u * c + 1u] = t3 + c;
          }
        } else {
          const uint32_t packed = ids[s] << kCornerBits;
          at[rank * 3u + 0] = packed | ((t3 + 0u) & kCornerMask);
          at[rank * 3u + 1] = packed | ((t3 + 1u) & kCornerMask);
          at[rank * 3u + 2] = packed | ((t3 + 2u) & kCornerMask);
        }
      }
      before += (uint32_t)__popcll((unsigned long long)m[h]);
    }
    filled += n3;
  }
  if (filled) flush();
}

// ------------------------------------------------------------------------------------------
// Triangle stage, test kernel (passes/cull_triangles.slang:27-90).  One wave per visible
// meshlet.  Each vertex is fetched, decoded and transformed once (lane = vertex) instead of
// once per referencing triangle corner; the triangle lanes then pick their three corners with
// ds_bpermute.  Same per-vertex arithmetic => same bits.  Result: a 64-bit pass mask per slot.
// ------------------------------------------------------------------------------------------
// WIDE (extension): up to 128 triangles per meshlet -> two 64-lane triangle passes and two
// 64-bit pass masks per slot (tri_masks[2*slot + half]).
// How the 16 slots of a wave are sequenced:
//  * the slot loop is fully unrolled and software-pipelined, so a wave keeps the index dwords of
//    kIdxAhead and the position gathers of kPosAhead meshlets in flight (bytes in flight are what bounds
//    this gather-heavy kernel) and every wait is a counted vmcnt(N) instead of the vmcnt(0) that a loop
//    back-edge or an exec-masked load block forces (the loop version of this kernel took 160 us where
//    this one takes 123 us on config 3);
//  * everything that is uniform per slot -- LOD pointers, the Meshlet record, projection_view * world --
//    travels through scalar loads into SGPRs (no v_readlane unpacking);
//  * no conditional loads: lanes beyond a meshlet's vertex / triangle count re-read its last element,
//    slots beyond the visible count re-do the last visible slot (their result is dropped), and a
//    meshlet with no vertices or triangles reads zero dwords of its own InstCache row;
//  * clip = ((m0*x + m1*y) + m2*z) + m3 as two packed pairs (xy, zw): same roundings, half the slots;
//  * the 16 pass masks are collected in lanes 0..15 (v_writelane) and stored once.
// vertex ids, micro indices and positions of a visible meshlet are read once per call: `nt` loads
// (measured on config 3: 140 -> 121 us per launch on the same box, whole frame +5 %)
// ... unless the geometry is SHARED between instances (CACHED instantiations, round 6): the engine's case -- a few meshes drawn many times
// (AssetManager_GLTF.cpp builds one blob per mesh, Scene.cpp:1248-1260 instances them).  There the `nt` hint keeps lines out of the caches that
// the next visible meshlet of another instance would have hit: the real-mesh frame (3 meshes, 8 700 instances) 0.729 -> 0.632 ms, fused
// kernels 164.5 / 360.6 -> 132.5 / 292.8 us with plain loads; on unique geometry plain loads cost what is quoted above.  The host picks
// (oxc_cull_geometry: mesh instances per Mesh record), oxc_debug_set_tuning(OXC_TUNE_TRI_LOADS) overrides.
#define OXC_TRI_LOAD_U32(base, index) (CACHED ? load_global_u32((base), (index)) : load_stream_u32((base), (index)))
#define OXC_TRI_LOAD_U2(base, index) (CACHED ? load_global_u2((base), (index)) : load_stream_u2((base), (index)))
// SMALL (extension, include/oxcull.h small_triangle_cull): after the two reference tests, drop a triangle whose
// screen-space bounding box covers no pixel centre.  The screen position is computed once per vertex (lane = vertex,
// two IEEE divisions) and fetched per corner like the clip coordinates; with SMALL off none of it is compiled in.
// FUSED (unordered_output, include/oxcull.h; tris_fused_body): the block also expands what it tested -- per work item (a span of kFusedTriSpan =
// 128 visible meshlets, two chunks, or a single chunk at the end of the launch) the pass masks stay in LDS, ONE returning atomic_add on
// DrawIndexedIndirect.index_count allocates the item's run of packed indices (cull_triangles.slang:71-88 does that per 64-thread workgroup; a
// single address retires ~88 atomics per microsecond here, hence the span), and the four waves write it the way tris_emit_body does.  No
// pass masks, chunk counts or visible ids go through memory and no emit launch follows; the runs land in arrival order (ascending inside an item).
// The two bodies of the triangle stage share the per-chunk pipeline (oxcull_tri_stages.inc / oxcull_tri_slots.inc):
//   tris_test_body  -- ordered form: pass masks + per-chunk counts to memory, k_cull_triangles_emit follows;
//   tris_fused_body -- unordered_output: the block also expands what it tested (FUSED above).
// Pipeline depths of the slot loop, measured on config 3 (us per launch): index loads 2 / position gathers 1 slot ahead -> 123, 3/2 -> 125 (SGPR
// spills), 6 waves/SIMD with 3/2 or 4/3 -> 139-143; the loop version: 160.  The WIDE instantiations use the same depths.
constexpr int kTriIdxAhead = 2, kTriPosAhead = 1;
// The corners of a triangle: nine ds_bpermute (36 B of LDS crossbar traffic per lane and pass) or, with the slot's transformed vertices written to
// LDS once, three 16-byte reads per pass (64 B).  T = 64 (one pass per slot): the reads lose outright, fused kernels 99 / 209 -> 112 / 266 us.  WIDE
// (two passes per slot, kernel bound by instruction issue rather than LDS bandwidth): 137 / 265 -> 132.5 / 258 us; 12-byte reads with the z flags
// kept as a ballot: 136.5 / 262.6.  So: LDS vertices for WIDE only.
template <bool LATE, bool WIDE, bool SMALL, bool CACHED = false>
OXC_DEV void tris_test_body(const TriTestArgs& a) {
  set_half_denorm_flush();
  constexpr int H = WIDE ? 2 : 1;
  constexpr int S = 16;  // slots per wave per chunk
  constexpr int kPosAhead = kTriPosAhead;  // position gathers (need the vertex ids) run this many slots ahead of the decision
  constexpr int kIdxAhead = kTriIdxAhead;  // vertex / micro index loads
  constexpr int kRecAhead = kIdxAhead + 1;      // scalar: Meshlet record
  constexpr int kRowAhead = kIdxAhead + 2;      // scalar: LOD pointers out of the InstCache row
  typedef const uint32_t __attribute__((address_space(4))) * k32;
  __shared__ uint32_t s_red[4];
  constexpr bool kLdsVerts = WIDE;
  __shared__ uint4 s_vert[kLdsVerts ? 4 : 1][kLdsVerts ? 64 : 1];  // per wave: the transformed vertices of the slot in hand (same-wave LDS traffic is in order)
  // (wave as a SCALAR in the CACHED instantiations: the compiler cannot see that threadIdx.x >> 6 is the same in all 64 lanes, and everything derived from it -- a
  //  slot's index, "is this slot inside the list" -- is computed per lane, ~6 VALU instructions per slot.  Round 6, measured: on shared geometry, where the kernel
  //  is 90 % VALU-busy, 131.7 / 293.7 -> 126.4 / 286.1 us; on unique geometry the extra scalar registers cost more than the instructions gave, 198.4 -> 201.4 us)
  const int lane = threadIdx.x & 63, wave = CACHED ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : (int)(threadIdx.x >> 6);
  const uint32_t V = a.tri_cmd[0];
  const uint32_t first = LATE ? a.vis[1] : 0u;  // cull_triangles.slang:34-37
  const uint32_t nchunks = (V + kTriChunk - 1) / kTriChunk;
  for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    // ---- lanes 0..15 fetch the MeshletInstance of this wave's 16 slots; everything that is uniform per
    // slot from there on (LOD pointers, Meshlet record, mvp) travels through scalar loads into SGPRs
    uint2 h_rec;
    {
      const uint32_t slot = min(chunk * kTriChunk + (uint32_t)(lane & 15) * 4 + wave, V - 1u);
      const uint32_t mli = a.visible[first + slot];
      h_rec = reinterpret_cast<const uint2*>(a.meshlet_instances)[mli];
    }
#include "oxcull_tri_stages.inc"
#pragma unroll
    for (int j = 0; j < kRowAhead; j++) stage_row(j);
#pragma unroll
    for (int j = 0; j < kRecAhead; j++) stage_rec(j);
#pragma unroll
    for (int j = 0; j < kIdxAhead; j++) stage_idx(j);
#pragma unroll
    for (int j = 0; j < kPosAhead; j++) stage_pos(j);
    uint32_t cnt = 0;
    uint32_t mlo[H], mhi[H];  // lane j: pass mask(s) of slot j
#pragma unroll
    for (int h = 0; h < H; h++) mlo[h] = mhi[h] = 0;
#define OXC_TRI_SLOT0 (chunk * kTriChunk)
#include "oxcull_tri_slots.inc"
#undef OXC_TRI_SLOT0
    {
      const uint32_t slot = chunk * kTriChunk + (uint32_t)lane * 4 + wave;
      if (lane < S && slot < V) {
#pragma unroll
        for (int h = 0; h < H; h++) a.tri_masks[(size_t)slot * H + h] = (uint64_t)mlo[h] | ((uint64_t)mhi[h] << 32);
      }
    }
    __syncthreads();
    if (lane == 0) s_red[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t c = s_red[0] + s_red[1] + s_red[2] + s_red[3];
      a.chunk_counts[chunk] = c;
      if (c) __hip_atomic_fetch_add(gptr(a.supers) + (chunk / kChunksPerSuper) * kSuperStride, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// Work distribution of the fused kernel (round 5).  A work item is a run of 64-slot chunks of the pass's visible list that one block tests,
// allocates with ONE atomic_add on index_count and expands: a SPAN (kFusedTriSpan slots, two chunks) or, at the end of the launch, a single
// chunk.  The spans of the first max(1, whole) rounds of the grid go by block index (span = block + round * grid); what is left of the
// last, partial round is handed out in CHUNKS, in arrival order, from kTriTicketCounters counters (counter x = block % K hands out the
// chunks congruent to x mod K: 2 048 blocks drawing at the end of their first span would queue for 26 us on ONE address).  The early
// launch of configs[2] is 2 524 spans on 2 048 resident blocks: with whole spans only, 476 blocks run a second span on a nearly empty
// machine.  Measured on the configs[2] frame (tools/kbench.py, profiles/r05_ab): fused kernels 98.5 / 205.6 -> 97.1 / 200.5 us, frame
// 511 -> 505 us with the emit launches in front; dynamic at span granularity gave half of it, a whole dynamic round more was worse.
constexpr uint32_t kTriTicketCounters = 16;  // (4 / 16 / 64 counters: the same frame time)
template <bool LATE, bool WIDE, bool SMALL, bool PAIR = false, bool CACHED = false>
OXC_DEV void tris_fused_body(const TriTestArgs& a) {
  static_assert(!PAIR || WIDE, "the pair form is the form of the 128-triangle meshlets");
  set_half_denorm_flush();
  constexpr int H = WIDE ? 2 : 1;
  constexpr int S = 16;  // slots per wave per chunk
  constexpr int kPosAhead = kTriPosAhead;
  constexpr int kIdxAhead = kTriIdxAhead;
  constexpr int kRecAhead = kIdxAhead + 1;
  constexpr int kRowAhead = kIdxAhead + 2;
  typedef const uint32_t __attribute__((address_space(4))) * k32;
  __shared__ uint32_t s_red[4];
  constexpr uint32_t kFSpan = kFusedTriSpan;                // visible meshlets per span (64, 128 or 256)
  constexpr uint32_t kChunksPerSpan = kFSpan / kTriChunk;
  static_assert(kFSpan == 64 || kFSpan == 128 || kFSpan == 256, "one slot per thread of the block at most, whole chunks");
  constexpr uint32_t kCornerBits = WIDE ? 9u : 8u;           // MESHLET_PRIMITIVE_BITS = 8 in the reference (visbuffer.slang:13)
  constexpr bool kLdsVerts = WIDE;
  constexpr bool kDynamic = kChunksPerSpan > 1u;  // (round 4: every span by block index)
  __shared__ uint4 s_vert[kLdsVerts ? 4 : 1][kLdsVerts ? 64 : 1];
  __shared__ uint32_t f_off[kFSpan];
  __shared__ uint64_t f_mask[kFSpan * H];
  __shared__ uint32_t f_id[kFSpan];
  __shared__ __attribute__((aligned(16))) uint32_t f_run[4 * (kEmitRun + 8u)];
  __shared__ uint32_t f_base;
  __shared__ uint32_t f_next;
  // (wave as a SCALAR in the CACHED instantiations: the compiler cannot see that threadIdx.x >> 6 is the same in all 64 lanes, and everything derived from it -- a
  //  slot's index, "is this slot inside the list" -- is computed per lane, ~6 VALU instructions per slot.  Round 6, measured: on shared geometry, where the kernel
  //  is 90 % VALU-busy, 131.7 / 293.7 -> 126.4 / 286.1 us; on unique geometry the extra scalar registers cost more than the instructions gave, 198.4 -> 201.4 us)
  const int lane = threadIdx.x & 63, wave = CACHED ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : (int)(threadIdx.x >> 6);
  const uint32_t V = a.tri_cmd[0];
  const uint32_t first = LATE ? a.vis[1] : 0u;  // cull_triangles.slang:34-37
  const uint32_t nchunks = (V + kTriChunk - 1) / kTriChunk;  // 64-slot chunks of the list (a chunk's slots beyond V re-do the last slot and leave empty masks)
  const uint32_t nspans = (nchunks + kChunksPerSpan - 1) / kChunksPerSpan;
  // chunks [0, static_chunks): whole spans by block index; [static_chunks, nchunks): single chunks by ticket
  const uint32_t static_chunks = kDynamic ? min(max(1u, nspans / gridDim.x) * gridDim.x * kChunksPerSpan, nchunks) : nchunks;
  const uint32_t Kc = max(1u, min(min(kTriTicketCounters, gridDim.x), a.ticket_count)), kx = blockIdx.x % Kc;
  uint32_t item = blockIdx.x * kChunksPerSpan;  // first chunk of the work item in hand ...
  uint32_t item_n = kChunksPerSpan;             // ... and its length in chunks (block-uniform)
  while (item < nchunks) {
    const uint32_t item_slot0 = item * kTriChunk;  // first slot of the item; LDS rows are indexed relative to it
    for (uint32_t c4 = 0; c4 < item_n; c4++) {
      const uint32_t chunk = item + c4;
      // ---- lanes 0..15 fetch the MeshletInstance of this wave's 16 slots; everything that is uniform per
      // slot from there on (LOD pointers, Meshlet record, mvp) travels through scalar loads into SGPRs
      uint2 h_rec;
      {
        const uint32_t slot = min(chunk * kTriChunk + (uint32_t)(lane & 15) * 4 + wave, V - 1u);
        const uint32_t mli = a.visible[first + slot];
        h_rec = reinterpret_cast<const uint2*>(a.meshlet_instances)[mli];
      }
#include "oxcull_tri_stages.inc"
#pragma unroll
      for (int j = 0; j < kRowAhead; j++) stage_row(j);
#pragma unroll
      for (int j = 0; j < kRecAhead; j++) stage_rec(j);
#pragma unroll
      for (int j = 0; j < kIdxAhead; j++) stage_idx(j);
#pragma unroll
      for (int j = 0; j < kPosAhead; j++) stage_pos(j);
      uint32_t cnt = 0;
      uint32_t mlo[H], mhi[H];  // lane j: pass mask(s) of slot j
#pragma unroll
      for (int h = 0; h < H; h++) mlo[h] = mhi[h] = 0;
#define OXC_TRI_SLOT0 (chunk * kTriChunk)
#include "oxcull_tri_slots.inc"
#undef OXC_TRI_SLOT0
      (void)cnt;
      if (lane < S) {  // slot (chunk c4, j = lane, wave) sits at c4 * 64 + j * 4 + wave of the item (a slot beyond V left an empty mask)
#pragma unroll
        for (int h = 0; h < H; h++) f_mask[(c4 * kTriChunk + (uint32_t)lane * 4 + wave) * H + h] = (uint64_t)mlo[h] | ((uint64_t)mhi[h] << 32);
      }
    }
    if (item_n < kChunksPerSpan && threadIdx.x >= item_n * kTriChunk && threadIdx.x < kFSpan) {  // a single chunk: the rest of the span's rows are empty
#pragma unroll
      for (int h = 0; h < H; h++) f_mask[threadIdx.x * H + h] = 0ull;
    }
    __syncthreads();
    // ---- the item is tested: allocate its run and expand it (tris_emit_body with the base taken from the counter itself)
    const uint32_t slot = item_slot0 + threadIdx.x;
    const bool in_span = threadIdx.x < kFSpan;
    uint32_t c = 0;
#pragma unroll
    for (int h = 0; h < H; h++) c += in_span ? (uint32_t)__popcll((unsigned long long)f_mask[(in_span ? threadIdx.x : 0u) * H + h]) : 0u;
    const uint32_t id = (threadIdx.x < item_n * kTriChunk && slot < V) ? a.visible[first + slot] : 0u;
    const uint32_t incl = wave_incl_scan(c, lane);
    if (lane == 63) s_red[wave] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (int k = 0; k < wave; k++) woff += s_red[k];
    if (in_span) {
      f_off[threadIdx.x] = woff + incl - c;
      f_id[threadIdx.x] = id;
    }
    // the block's next item: the span one grid further while that is a static one, a drawn chunk after that
    const uint32_t next_static = item + gridDim.x * kChunksPerSpan;
    const bool draw = kDynamic && static_chunks < nchunks && (item_n < kChunksPerSpan || next_static >= static_chunks);  // (block-uniform)
    if (threadIdx.x == 255) {
      const uint32_t total3 = (woff + incl) * 3u;
      uint32_t base = 0u, tk = 0u;
      if (total3) base = __hip_atomic_fetch_add(gptr(a.draw_cmd), total3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // DrawIndexedIndirect.index_count
      // (pairs have no id limit, so a call CAN be asked for more than the 2^32 - 1 indices a VkDrawIndexedIndirectCommand counts: the run that wraps the
      //  counter zeroes instanceCount -- the draw becomes a no-op and oxc_read_counters reports it; include/oxcull.h, wide_triangle_index = 2)
      if (PAIR && base + total3 < base) gptr(a.draw_cmd)[1] = 0u;
      if (draw) tk = __hip_atomic_fetch_add(gptr(a.ticket) + kx * kSuperStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (both round trips overlap)
      f_next = tk;
      f_base = base;
    }
    __syncthreads();
    {  // every wave expands a quarter of the ITEM's slots (a drawn chunk is 64 slots: 16 per wave instead of 32 for two waves and none for
       // the other two: early launch 99.7 -> 96.5 us, frame -3.5 us)
      const int per_wave = (int)(item_n * kTriChunk / 4u);
      if (PAIR)
        expand_slots_wide<H, kCornerBits, true>(f_mask, f_id, wave * per_wave, per_wave, ((uint64_t)f_base + (uint64_t)f_off[wave * per_wave] * 3u) * 2u, a.out, f_run + wave * (kEmitRun + 8u), lane);
      else
        expand_slots_wide<H, kCornerBits, false>(f_mask, f_id, wave * per_wave, per_wave, f_base + f_off[wave * per_wave] * 3u, a.out, f_run + wave * (kEmitRun + 8u), lane);
    }
    const uint32_t tk = f_next;
    __syncthreads();  // the item's LDS rows (and f_next) are rewritten by the block's next item
    if (draw) {
      item = static_chunks + tk * Kc + kx;
      item_n = 1u;
    } else {
      item = (kDynamic && next_static >= static_chunks) ? nchunks : next_static;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Triangle stage, emit kernel: ordered expansion of the pass masks into packed indices
// (visbuffer.slang:13-14, cull_triangles.slang:82-88) and DrawIndexedIndirect.index_count.
// ------------------------------------------------------------------------------------------
template <bool LATE, bool WIDE, bool PAIR = false>
OXC_DEV void tris_emit_body(const TriEmitArgs& a) {
  static_assert(!PAIR || WIDE, "the pair form is the form of the 128-triangle meshlets");
  constexpr int H = WIDE ? 2 : 1;
  constexpr uint32_t kCornerBits = WIDE ? 9u : 8u;  // MESHLET_PRIMITIVE_BITS = 8 in the reference (visbuffer.slang:13)
  __shared__ uint32_t s_red[4];
  __shared__ uint32_t s_wave[4];
  __shared__ uint32_t s_off[256];
  __shared__ uint64_t s_mask[256 * H];
  __shared__ uint32_t s_id[256];
  __shared__ __attribute__((aligned(16))) uint32_t s_run[4 * (kEmitRun + 8u)];
  static_assert(kEmitRun >= 384u + 3u && kEmitRun % 4u == 0u, "a run holds at least one WIDE slot; rows stay 16-byte aligned");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t V = a.tri_cmd[0];
  const uint32_t first = LATE ? a.vis[1] : 0u;
  const uint32_t nspans = (V + kTriSpan - 1) / kTriSpan;
  constexpr uint32_t kChunksPerSpan = kTriSpan / kTriChunk;
  for (uint32_t span = blockIdx.x; span < nspans; span += gridDim.x) {
    const uint32_t base = chunk_base_256(a.supers, a.chunk_counts, span * kChunksPerSpan, s_red);
    const uint32_t slot = span * kTriSpan + threadIdx.x;
    uint64_t mask[H];
    uint32_t id = 0;
    uint32_t c = 0;
#pragma unroll
    for (int h = 0; h < H; h++) {
      mask[h] = slot < V ? a.tri_masks[(size_t)slot * H + h] : 0ull;
      c += (uint32_t)__popcll((unsigned long long)mask[h]);
    }
    if (slot < V) id = a.visible[first + slot];
    const uint32_t incl = wave_incl_scan(c, lane);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (int k = 0; k < wave; k++) woff += s_wave[k];
    s_off[threadIdx.x] = woff + incl - c;
#pragma unroll
    for (int h = 0; h < H; h++) s_mask[threadIdx.x * H + h] = mask[h];
    s_id[threadIdx.x] = id;
    if (threadIdx.x == 255 && span == nspans - 1) {
      a.draw_cmd[0] = (base + woff + incl) * 3u;  // DrawIndexedIndirect.index_count
      if (PAIR && (uint64_t)(base + woff + incl) * 3u > 0xFFFFFFFFull) a.draw_cmd[1] = 0u;  // (more indices than the command can count: see tris_fused_body)
    }
    __syncthreads();
    // each wave expands its 64 slots: one contiguous run of the index list (expand_slots_wide above)
    if (PAIR)
      expand_slots_wide<H, kCornerBits, true>(s_mask, s_id, wave * 64, 64, (uint64_t)(base + s_off[wave * 64]) * 6u, a.out, s_run + wave * (kEmitRun + 8u), lane);
    else
      expand_slots_wide<H, kCornerBits, false>(s_mask, s_id, wave * 64, 64, (base + s_off[wave * 64]) * 3u, a.out, s_run + wave * (kEmitRun + 8u), lane);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// HiZ (passes/hiz.slang).  The pyramid is a pure function of the depth image, so the
// decomposition is free: one 256-thread block builds mips 0..6 of a 64x64 mip-0 tile through
// registers + LDS; a single-block tail kernel finishes the remaining mips from LDS.
// mip 0 is the reference's NEAREST point sample at uv=(texel+1)/extent (hiz.slang:92-95).
// ------------------------------------------------------------------------------------------
OXC_DEV float hiz_point_sample(const float* __restrict__ depth, uint32_t dw, uint32_t dh, uint32_t x, uint32_t y, float invx,
                               float invy) {
  float uu = (float)x * invx + invx;
  float vv = (float)y * invy + invy;
  int32_t sx = cvt_i32_sat(floorf(uu * (float)dw));
  int32_t sy = cvt_i32_sat(floorf(vv * (float)dh));
  sx = min(max(sx, 0), (int32_t)dw - 1);
  sy = min(max(sy, 0), (int32_t)dh - 1);
  return depth[(size_t)sy * dw + sx];  // (`nt` LOADS here are slower: 80 -> 84 us in round 1, 46 -> 71 us with the nt stores of round 4)
}

__global__ __launch_bounds__(256) void k_hiz_tile(HizArgs a) {
  __shared__ float s_a[16 * 16];
  __shared__ float s_b[8 * 8];
  const uint32_t tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const uint32_t W = a.w, H = a.h;
  const uint32_t x0 = blockIdx.x * 64 + tx * 4, y0 = blockIdx.y * 64 + ty * 4;
  const float invx = 1.0f / (float)W, invy = 1.0f / (float)H;  // hiz.slang:33
  float m[4][4];
#pragma unroll
  for (int r = 0; r < 4; r++)
#pragma unroll
    for (int c = 0; c < 4; c++) m[r][c] = hiz_point_sample(a.depth, a.dw, a.dh, x0 + c, y0 + r, invx, invy);
  float* mip0 = a.hiz + a.level_off[0];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    float4 v = make_float4(m[r][0], m[r][1], m[r][2], m[r][3]);
    // `nt`: mip 0 is 3/4 of the pyramid (64 MB at 4096^2) and nothing reads it back soon -- the meshlet tests sample a few texels of
    // it per candidate.  Written with the default policy its lines sat in L2 / the Infinity Cache and their write-back was paid by
    // the kernels that followed (configs[2] frame, us: k_hiz_tile 74.6 -> 62.3, triangle tests 80.7 / 160.2 -> 79.1 / 153.2, frame
    // 565 -> 544; byte-identical).  Kept: nt for mip 0 only (nt for mip 1 too measured no better).
    {
      typedef float f4v __attribute__((ext_vector_type(4)));
      __builtin_nontemporal_store(f4v{v.x, v.y, v.z, v.w}, reinterpret_cast<f4v*>(mip0 + (size_t)(y0 + r) * W + x0));
    }
  }
  if (a.levels <= 1) return;
  // mip 1: 2x2 per thread
  float q[2][2];
#pragma unroll
  for (int r = 0; r < 2; r++)
#pragma unroll
    for (int c = 0; c < 2; c++)
      q[r][c] = fminf(fminf(m[2 * r][2 * c], m[2 * r][2 * c + 1]), fminf(m[2 * r + 1][2 * c], m[2 * r + 1][2 * c + 1]));
  {
    float* mip1 = a.hiz + a.level_off[1];
    const uint32_t w1 = W >> 1;
#pragma unroll
    for (int r = 0; r < 2; r++) {
      *reinterpret_cast<float2*>(mip1 + (size_t)((y0 >> 1) + r) * w1 + (x0 >> 1)) = make_float2(q[r][0], q[r][1]);
    }
  }
  if (a.levels <= 2) return;
  // mip 2: one per thread
  float d = fminf(fminf(q[0][0], q[0][1]), fminf(q[1][0], q[1][1]));
  (a.hiz + a.level_off[2])[(size_t)(y0 >> 2) * (W >> 2) + (x0 >> 2)] = d;
  s_a[ty * 16 + tx] = d;
  if (a.levels <= 3) return;
  __syncthreads();
  // mip 3: 8x8 per tile
  if (threadIdx.x < 64) {
    uint32_t px = threadIdx.x & 7, py = threadIdx.x >> 3;
    float v = fminf(fminf(s_a[(2 * py) * 16 + 2 * px], s_a[(2 * py) * 16 + 2 * px + 1]),
                    fminf(s_a[(2 * py + 1) * 16 + 2 * px], s_a[(2 * py + 1) * 16 + 2 * px + 1]));
    (a.hiz + a.level_off[3])[(size_t)(blockIdx.y * 8 + py) * (W >> 3) + blockIdx.x * 8 + px] = v;
    s_b[py * 8 + px] = v;
  }
  if (a.levels <= 4) return;
  __syncthreads();
  // mip 4: 4x4
  if (threadIdx.x < 16) {
    uint32_t px = threadIdx.x & 3, py = threadIdx.x >> 2;
    float v = fminf(fminf(s_b[(2 * py) * 8 + 2 * px], s_b[(2 * py) * 8 + 2 * px + 1]),
                    fminf(s_b[(2 * py + 1) * 8 + 2 * px], s_b[(2 * py + 1) * 8 + 2 * px + 1]));
    (a.hiz + a.level_off[4])[(size_t)(blockIdx.y * 4 + py) * (W >> 4) + blockIdx.x * 4 + px] = v;
    s_a[py * 4 + px] = v;
  }
  if (a.levels <= 5) return;
  __syncthreads();
  // mip 5: 2x2, mip 6: 1
  if (threadIdx.x < 4) {
    uint32_t px = threadIdx.x & 1, py = threadIdx.x >> 1;
    float v = fminf(fminf(s_a[(2 * py) * 4 + 2 * px], s_a[(2 * py) * 4 + 2 * px + 1]),
                    fminf(s_a[(2 * py + 1) * 4 + 2 * px], s_a[(2 * py + 1) * 4 + 2 * px + 1]));
    (a.hiz + a.level_off[5])[(size_t)(blockIdx.y * 2 + py) * (W >> 5) + blockIdx.x * 2 + px] = v;
    s_b[py * 2 + px] = v;
  }
  if (a.levels <= 6) return;
  __syncthreads();
  if (threadIdx.x == 0) {
    float v = fminf(fminf(s_b[0], s_b[1]), fminf(s_b[2], s_b[3]));
    (a.hiz + a.level_off[6])[(size_t)blockIdx.y * (W >> 6) + blockIdx.x] = v;
  }
}

// Generic mip 0 for pyramids smaller than one tile.
__global__ __launch_bounds__(256) void k_hiz_mip0_generic(HizArgs a) {
  const uint32_t n = a.w * a.h;
  const float invx = 1.0f / (float)a.w, invy = 1.0f / (float)a.h;
  float* mip0 = a.hiz + a.level_off[0];
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    uint32_t x = i % a.w, y = i / a.w;
    mip0[i] = hiz_point_sample(a.depth, a.dw, a.dh, x, y, invx, invy);
  }
}

// Tail: levels (start+1 .. levels-1) from level `start`, which must hold <= 4096 texels.
// 2x2 min with edge clamp (max(1, dim>>k) sizing).
__global__ __launch_bounds__(1024) void k_hiz_tail(HizArgs a, uint32_t start) {
  __shared__ float s_buf[2][4096];
  uint32_t pw = mip_dim(a.w, start), ph = mip_dim(a.h, start);
  const float* src = a.hiz + a.level_off[start];
  for (uint32_t i = threadIdx.x; i < pw * ph; i += blockDim.x) s_buf[0][i] = src[i];
  __syncthreads();
  int cur = 0;
  for (uint32_t k = start + 1; k < a.levels; k++) {
    uint32_t cw = mip_dim(a.w, k), ch = mip_dim(a.h, k);
    float* dst = a.hiz + a.level_off[k];
    for (uint32_t i = threadIdx.x; i < cw * ch; i += blockDim.x) {
      uint32_t x = i % cw, y = i / cw;
      uint32_t xa = 2 * x, ya = 2 * y;
      uint32_t xb = xa + 1 < pw ? xa + 1 : pw - 1, yb = ya + 1 < ph ? ya + 1 : ph - 1;
      float v = fminf(fminf(s_buf[cur][ya * pw + xa], s_buf[cur][ya * pw + xb]),
                      fminf(s_buf[cur][yb * pw + xa], s_buf[cur][yb * pw + xb]));
      s_buf[cur ^ 1][i] = v;
      dst[i] = v;
    }
    __syncthreads();
    cur ^= 1;
    pw = cw;
    ph = ch;
  }
}

// ------------------------------------------------------------------------------------------
// misc
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_seed_slot(uint32_t* slot, uint32_t total) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    slot[SLOT_VIS + 0] = total;
    slot[SLOT_VIS + 1] = 0;
    slot[SLOT_VIS + 2] = 0;
    slot[SLOT_MESHLETS_CMD + 0] = (total + 63u) / 64u;
    slot[SLOT_MESHLETS_CMD + 1] = 1;
    slot[SLOT_MESHLETS_CMD + 2] = 1;
  }
}

__global__ __launch_bounds__(64) void k_pack_counters(const uint32_t* vis, const uint32_t* tri_cmd, const uint32_t* draw_cmd, uint32_t* out4) {
  if (threadIdx.x == 0) {
    out4[0] = tri_cmd ? tri_cmd[0] : 0u;
    out4[1] = vis ? vis[1] : 0u;
    out4[2] = vis ? vis[2] : 0u;
    out4[3] = draw_cmd ? draw_cmd[0] : 0u;
  }
}

__global__ __launch_bounds__(256) void k_stream_read(const uint4* __restrict__ p, uint64_t n16, uint32_t* sink) {
  uint32_t acc = 0;
  // one contiguous 16 KiB tile per block iteration: four independent 16 B loads in flight per lane
  const uint64_t tiles = n16 / 1024;
  for (uint64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const uint4* q = p + t * 1024 + threadIdx.x;
    uint4 a = q[0], b = q[256], c = q[512], d = q[768];
    acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
  }
  for (uint64_t i = tiles * 1024 + (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) {
    uint4 v = p[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x9E3779B9u) *sink = acc;  // keep the loads alive
}

__global__ __launch_bounds__(256) void k_debug_decode_bounds(const uint4* __restrict__ bounds, uint32_t n, float* __restrict__ out10) {
  set_half_denorm_flush();
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint4 b = bounds[i];
    float* o = out10 + (size_t)i * 10;
    o[0] = dequantize_half(b.x & 0xFFFFu);
    o[1] = dequantize_half(b.x >> 16);
    o[2] = dequantize_half(b.y & 0xFFFFu);
    o[3] = dequantize_half(b.z & 0xFFFFu);
    o[4] = dequantize_half(b.z >> 16);
    o[5] = dequantize_half(b.w & 0xFFFFu);
    o[6] = s8_over_127((int32_t)(b.y << 8) >> 24);
    o[7] = s8_over_127((int32_t)b.y >> 24);
    o[8] = s8_over_127((int32_t)(b.w << 8) >> 24);
    o[9] = s8_over_127((int32_t)b.w >> 24);
  }
}

// ------------------------------------------------------------------------------------------
// kernel entry points: single-frame wrappers and batched wrappers (blockIdx.y = batch element)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prepare_instances(PrepareArgs a) { prepare_body(a, blockIdx.y); }
__global__ __launch_bounds__(1024) void k_scan_mesh_counts(ScanArgs a) { scan_body(a); }
__global__ __launch_bounds__(256) void k_expand_meshlet_instances(ExpandArgs a) { expand_body(a); }
// (Capping SGPRs at 80 for 8 waves/SIMD -- the compiler otherwise keeps ~106 live -- was measured:
// plain kernel 36.7 -> 38.8 us per 4M meshlets, HiZ variant 183 -> 175 us; not kept.)
// (The occlusion variants sit at 96-98 VGPRs: the second bound keeps them at the 5 waves per SIMD they have always run with.)
template <bool HIZ, bool OCCL, bool LATE, int G = (int)kGroupsPerWave>
__global__ __launch_bounds__(1024 / G, (HIZ && (OCCL || LATE)) ? 5 : 1) void k_cull_meshlets_test(MeshletTestArgs a) {
  if constexpr (!HIZ)
    meshlets_plain_body<G>(a);
  else
    meshlets_hiz_body<OCCL, LATE, G>(a);
}
// unordered_output: the plain body appending its survivors itself (MeshletTestArgs::out)
__global__ __launch_bounds__(64 * kUnordBlockWaves) void k_cull_meshlets_test_unordered(MeshletTestArgs a) { meshlets_plain_body<(int)kGroupsPerWave, true, kUnordBlockWaves>(a); }
// the two calls of a frame sharing the frustum test (MeshletTestArgs::share)
template <bool LATE>
__global__ __launch_bounds__(1024 / kHizGroupsPerWave, 5) void k_cull_meshlets_test_shared(MeshletTestArgs a) {
  // (Round 4 built the form in which a wave takes several consecutive steps and fills its 64-lane occlusion batches ACROSS them -- a tagged
  // per-wave LDS queue, results as cleared bits in per-(step, group) words, steps finished from those words after the last flush; byte-
  // identical on the share / fuzz / full-size tests -- and measured it: 135 / 124 us (2 steps per item, 5 waves per SIMD, 30 VGPRs
  // spilled), 96 / 78 us at 4 waves per SIMD, 101 / 108 us with one step per item (the queue machinery alone), against 78 / 65 us for
  // this kernel.  Fuller batches (the early call's run at 31 % of the lanes, the late call's at 66 %) do not pay for the deferred finish,
  // the LDS words and a third inlined copy of the batch.  Removed: git log -S meshlets_run_body.)
  meshlets_hiz_body<true, LATE, (int)kHizGroupsPerWave, LATE ? 2 : 1>(a);
}
// measurement aid: the occlusion kernels that also count the candidates reaching test_occlusion (MeshletTestArgs::dbg_occlusion)
template <bool LATE, int SHARE>
__global__ __launch_bounds__(1024 / kHizGroupsPerWave, 4) void k_cull_meshlets_test_counting(MeshletTestArgs a) {
  meshlets_hiz_body<true, LATE, (int)kHizGroupsPerWave, SHARE, true>(a);
}
template <bool HIZ, bool LATE>
__global__ __launch_bounds__(256) void k_cull_meshlets_emit(MeshletEmitArgs a) {
  meshlets_emit_body<HIZ, LATE>(a);
}
constexpr int kTriWaves = 8;
constexpr int kTriWideWaves = 4;  // (waves per SIMD of the WIDE instantiations: at 6 the 80-VGPR budget spills 25-36 VGPRs to scratch -- 113 / 234 us per launch on the 8 M x 124-triangle frame; 5: 99 / 211; 4: 90 / 185; 3: the same)
template <bool LATE, bool WIDE, bool SMALL>
__global__ __launch_bounds__(256, WIDE ? kTriWideWaves : (SMALL ? 6 : kTriWaves)) void k_cull_triangles_test(TriTestArgs a) {
  tris_test_body<LATE, WIDE, SMALL>(a);
}
template <bool LATE, bool WIDE>
__global__ __launch_bounds__(256) void k_cull_triangles_emit(TriEmitArgs a) {
  tris_emit_body<LATE, WIDE>(a);
}
template <bool LATE, bool WIDE, bool SMALL>
__global__ __launch_bounds__(256, WIDE ? kTriWideWaves : (SMALL ? 6 : kTriWaves)) void k_cull_triangles_fused(TriTestArgs a) {
  tris_fused_body<LATE, WIDE, SMALL>(a);
}
// wide_triangle_index = 2: the WIDE instantiations writing {id, corner} pairs (kernels of their own name: the profiles key on the names above)
template <bool LATE, bool SMALL>
__global__ __launch_bounds__(256, kTriWideWaves) void k_cull_triangles_fused_pairs(TriTestArgs a) {
  tris_fused_body<LATE, true, SMALL, true>(a);
}
template <bool LATE>
__global__ __launch_bounds__(256) void k_cull_triangles_emit_pairs(TriEmitArgs a) {
  tris_emit_body<LATE, true, true>(a);
}
// shared (instanced) geometry: plain loads instead of `nt` for vertex ids, micro indices and positions (OXC_TRI_LOAD_*); INDEX = wide_triangle_index
template <bool LATE, int INDEX>
// (waves per SIMD of the T = 64 instantiation on the real-mesh frame: 8 -> 0.620 ms, 7 -> 0.628 although its 94 SGPRs spill nothing, 6 -> 0.79)
__global__ __launch_bounds__(256, INDEX ? kTriWideWaves : kTriWaves) void k_cull_triangles_fused_cached(TriTestArgs a) {
  tris_fused_body<LATE, INDEX != 0, false, INDEX == 2, true>(a);
}
template <bool LATE, bool WIDE>
__global__ __launch_bounds__(256, WIDE ? kTriWideWaves : kTriWaves) void k_cull_triangles_test_cached(TriTestArgs a) {
  tris_test_body<LATE, WIDE, false, true>(a);
}

// Batched prepare: gets every element's core by value (kernarg), rebuilds the per-stage argument blocks of its
// element in device memory for the later kernels of the call (expand_batch_core), and prepares that element's
// instance rows.  The switch keeps the kernarg indexing static (a dynamic index into a by-value aggregate
// would be lowered through scratch).
__global__ __launch_bounds__(256) void k_prepare_batch(BatchBlob blob, BatchElem* __restrict__ dev) {
#define OXC_PREPARE_CASE(i)                                                                             \
  case i: {                                                                                             \
    if (blockIdx.x == 0 && threadIdx.x == 0) expand_batch_core(blob.core[i], dev[blob.first + i]);      \
    PrepareArgs pa;                                                                                     \
    prepare_args_of(blob.core[i], pa);                                                                  \
    prepare_body(pa, 0u);                                                                               \
  } break;
  switch (blockIdx.y) {
    OXC_PREPARE_CASE(0)
    OXC_PREPARE_CASE(1)
    OXC_PREPARE_CASE(2)
    OXC_PREPARE_CASE(3)
    OXC_PREPARE_CASE(4)
    OXC_PREPARE_CASE(5)
    OXC_PREPARE_CASE(6)
    OXC_PREPARE_CASE(7)
    OXC_PREPARE_CASE(8)
    OXC_PREPARE_CASE(9)
    OXC_PREPARE_CASE(10)
    OXC_PREPARE_CASE(11)
    OXC_PREPARE_CASE(12)
    OXC_PREPARE_CASE(13)
    OXC_PREPARE_CASE(14)
    default: {
      if (blockIdx.x == 0 && threadIdx.x == 0) expand_batch_core(blob.core[15], dev[blob.first + 15]);
      PrepareArgs pa;
      prepare_args_of(blob.core[15], pa);
      prepare_body(pa, 0u);
    } break;
  }
#undef OXC_PREPARE_CASE
  static_assert(kBatchPerPrepare == 16, "one case per element core of the blob");
}
__global__ __launch_bounds__(1024) void k_scan_batch(const BatchElem* __restrict__ dev) { scan_body(dev[blockIdx.y].scan); }
__global__ __launch_bounds__(256) void k_expand_batch(const BatchElem* __restrict__ dev) { expand_body(dev[blockIdx.y].expand); }
__global__ __launch_bounds__(64 * kPlainBlockWaves) void k_cull_meshlets_test_batch(const BatchElem* __restrict__ dev) {
  meshlets_plain_body<(int)kGroupsPerWave>(dev[blockIdx.y].test);
}
__global__ __launch_bounds__(256) void k_cull_meshlets_emit_batch(const BatchElem* __restrict__ dev) {
  meshlets_emit_body<false, false>(dev[blockIdx.y].emit);
}
__global__ __launch_bounds__(256, 8) void k_cull_triangles_test_batch(const BatchElem* __restrict__ dev) {
  tris_test_body<false, false, false>(dev[blockIdx.y].ttest);
}
__global__ __launch_bounds__(256) void k_cull_triangles_emit_batch(const BatchElem* __restrict__ dev) {
  tris_emit_body<false, false>(dev[blockIdx.y].temit);
}

// ------------------------------------------------------------------------------------------
// Multi-view meshlet stage (oxcull_kernels.hpp, MvArgs): several views of ONE scene in one pass.
// ------------------------------------------------------------------------------------------
constexpr uint32_t kMvChunk = 256;  // meshlets per wave step: 4 groups of 64

// (1) per mesh instance: group the views, count chunks.  Sixteen lanes per instance, lane = view.  Also publishes the view table to
// device memory and zeroes the counters the later kernels accumulate into.
__global__ __launch_bounds__(256) void k_mv_group(MvArgs a, MvBlob blob) {
  static_assert(kMaxBatch == 16, "one lane per view in a 16-lane segment");
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, nthreads = gridDim.x * blockDim.x;
  const int lane = threadIdx.x & 63;
  const uint32_t v = (uint32_t)lane & 15u, seg = (uint32_t)lane & ~15u;
  const MvView* const table = blob.v;  // (kernarg segment: indexed by lane)
  if (tid < a.views) a.dev[tid] = table[tid];
  for (uint32_t u = 0; u < a.views; u++)
    for (uint32_t i = tid; i < table[u].n_supers; i += nthreads) table[u].supers[i * kSuperStride] = 0;
  if (tid < kTicketCounters) a.tickets[tid * kSuperStride] = 0;
  const uint32_t per_round = nthreads >> 4, rounds = (a.M + per_round - 1) / per_round;
  for (uint32_t r = 0; r < rounds; r++) {
    const uint32_t mi = r * per_round + (tid >> 4);
    const bool live = mi < a.M && v < a.views;
    uint64_t kb = 0;
    uint32_t kt = 0, cnt = 0;
    if (live) {
      const MvView& w = table[v];
      const uint32_t c = w.mesh_counts[mi], off = w.mesh_offsets[mi];
      cnt = off >= w.n_cap ? 0u : min(c, w.n_cap - off);  // records beyond the view's list capacity do not exist (expand_body)
      const InstCache* row = w.rows + mi;
      kb = row->bounds;
      kt = row->transform_index;
      w.vchunks[mi] = (cnt + kMvChunk - 1) / kMvChunk;
      if (w.runs) {  // the view's list as runs (implicit_meshlet_instances): record i = {mi, i - first} for first <= i < first + count
        w.runs[2 * mi] = cnt ? off : 0u;
        w.runs[2 * mi + 1] = cnt;
      }
    }
    uint32_t same = 0;  // views of this instance with my key
#pragma unroll
    for (uint32_t u = 0; u < 16; u++) {
      const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)kb, (int)(seg + u), 64), hi = (uint32_t)__shfl((int)(uint32_t)(kb >> 32), (int)(seg + u), 64);
      const uint32_t tu = (uint32_t)__shfl((int)kt, (int)(seg + u), 64), cu = (uint32_t)__shfl((int)cnt, (int)(seg + u), 64);
      if (cnt != 0u && cu == cnt && tu == kt && (((uint64_t)hi << 32) | lo) == kb) same |= 1u << u;
    }
    const bool leader = cnt != 0u && (uint32_t)__builtin_ctz(same | 0x10000u) == v;
    // Views of the group whose frustum plane NORMALS are the leader's, bit for bit (the cascades of one light: orthographic matrices that
    // differ by scale and translation): k_mv_test computes a box's six plane distances once for all of them (frustum_plane_dots).  The
    // signs derive from the same sign bits.  Bits 16..31 of the group's view_mask.
    bool eq = cnt != 0u;
    if (eq) {  // my row's first 96 bytes against the leader's (six 16-byte loads each; the leader's are the same lines for the whole group)
      const uint32_t lead = (uint32_t)__builtin_ctz(same | 0x10000u) & 15u;
      const uint4* mine = reinterpret_cast<const uint4*>(table[v].rows + mi);
      const uint4* theirs = reinterpret_cast<const uint4*>(table[lead].rows + mi);
#pragma unroll
      for (int q = 0; q < 6; q++) {  // planes2[p][c][k] at dword p * 8 + c * 2 + k: the normals are c < 3, i.e. all of the even quads and .xy of the odd ones
        const uint4 x = mine[q], y = theirs[q];
        eq = eq && x.x == y.x && x.y == y.y && ((q & 1) || (x.z == y.z && x.w == y.w));
      }
    }
    const uint32_t nsame = (uint32_t)(__ballot(eq) >> seg) & same & 0xFFFFu;
    const uint32_t leaders = (uint32_t)(__ballot(leader) >> seg) & 0xFFFFu;
    const uint32_t g = (uint32_t)__popc(leaders & ((1u << v) - 1u)), ngroups = (uint32_t)__popc(leaders);
    uint32_t chunks = leader ? (cnt + kMvChunk - 1) / kMvChunk : 0u;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) chunks += (uint32_t)__shfl_xor((int)chunks, o, 64);
    if (mi < a.M) {
      MvGroup* out = a.groups + (size_t)mi * a.views;
      if (leader) out[g] = MvGroup{kb, cnt, same | (nsame << 16)};
      if (v >= ngroups && v < a.views) out[v] = MvGroup{0, 0, 0};
      if (v == 0) a.grp_chunks[mi] = chunks;
    }
  }
}

// (2) exclusive prefixes: per view over vchunks (the view's chunk numbering), and over grp_chunks (the step list)
__global__ __launch_bounds__(1024) void k_mv_scan(MvArgs a) {
  ScanArgs sa;
  if (blockIdx.y < a.views) {
    const MvView& w = a.dev[blockIdx.y];
    sa = ScanArgs{w.vchunks, w.vchunk0, a.M, 0xFFFFFFFFu, w.scan_total, w.scan_total + 1};
  } else {
    sa = ScanArgs{a.grp_chunks, a.inst_step0, a.M, 0xFFFFFFFFu, a.step_total, a.step_total + 1};
  }
  scan_body(sa);
}

// (3) the step list: one wave per mesh instance writes {instance, group | chunk << 8} for every chunk of every group
__global__ __launch_bounds__(256) void k_mv_steps(MvArgs a) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t mi = wave; mi < a.M; mi += nwaves) {
    uint32_t at = a.inst_step0[mi];
    const MvGroup* g = a.groups + (size_t)mi * a.views;
    for (uint32_t k = 0; k < a.views; k++) {
      const uint32_t count = g[k].count;
      if (count == 0u) break;
      const uint32_t chunks = (count + kMvChunk - 1) / kMvChunk;
      for (uint32_t c = lane; c < chunks; c += 64) reinterpret_cast<uint32_t*>(a.steps)[(size_t)(at + c) * 2] = mi, reinterpret_cast<uint32_t*>(a.steps)[(size_t)(at + c) * 2 + 1] = k | (c << 8);
      at += chunks;
    }
  }
}

// (4) the test: cull_meshlets.slang:23-73 for every view of the step's group over one load of the bounds records.
// SAME_POS: every view has the same camera position -- the normal cone (world matrix, normal matrix and scale are the group's: same
// transform) is then evaluated once per meshlet and kept as a 64-bit lane mask per 64-meshlet group; otherwise once per view.
template <bool SAME_POS>
// (waves per SIMD, same camera position, 10 M meshlets x 16 views: 6 -> 141 us, 7 -> 138, 8 -> 130)
__global__ __launch_bounds__(256, SAME_POS ? 6 : 4) void k_mv_test(MvArgs a) {
  set_half_denorm_flush();
  constexpr int G = 4;
  const int lane = threadIdx.x & 63;
  const uint32_t nsteps = gptr(a.step_total)[0];
  const uint32_t K = min(kTicketCounters, gridDim.x), kx = blockIdx.x % K;
  uint32_t* const ticket = a.tickets + kx * kSuperStride;
  auto draw_ticket = [&]() -> uint32_t {
    uint32_t t = 0;
    if (lane == 0) t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return t;
  };
  uint32_t step = OXC_TICKET_STEP(readlane_u(draw_ticket(), 0), K, kx);
  while (step < nsteps) {
    const uint32_t next_ticket = draw_ticket();
    const uint2 sd = load_global_u2(reinterpret_cast<uint64_t>(a.steps), step);
    const uint32_t mi = readfirst_u(sd.x), gi = readfirst_u(sd.y) & 0xFFu, c = readfirst_u(sd.y) >> 8;
    const kconst32p grp = (kconst32p)(reinterpret_cast<uint64_t>(a.groups + (size_t)mi * a.views + gi));
    const uint64_t bounds = (uint64_t)grp[0] | ((uint64_t)grp[1] << 32);
    const uint32_t count = grp[2], vmask = grp[3] & 0xFFFFu, nmask = grp[3] >> 16;  // nmask: views with the leader's plane normals (k_mv_group)
    const uint32_t k0 = c * kMvChunk;
    uint4 bnd[G];
#pragma unroll
    for (int j = 0; j < G; j++) bnd[j] = load_stream_u4(bounds, min(k0 + (uint32_t)j * 64u + (uint32_t)lane, count - 1u));
    // Lane v holds what is view v's own in this step -- the six plane offsets of its row, the chunk's number in the view's numbering, the list
    // index of the chunk's first meshlet: one vector-load chain per step (table entry -> row / prefix arrays) instead of a scalar round trip
    // per view inside the loop below.
    float lw[6];
    uint32_t lp, lid0;
    {
      const MvView* wl = a.dev + ((uint32_t)lane < a.views ? (uint32_t)lane : 0u);
      const uint32_t* rl = reinterpret_cast<const uint32_t*>(wl->rows + mi);
#pragma unroll
      for (int p = 0; p < 3; p++) {
        lw[2 * p] = -asf(rl[kRowPlanes + p * 8 + 6]);
        lw[2 * p + 1] = -asf(rl[kRowPlanes + p * 8 + 7]);
      }
      lp = wl->vchunk0[mi] + c;              // this chunk in the view's own numbering
      lid0 = wl->mesh_offsets[mi] + k0;
    }
#define OXC_MV_DECODE(b)                                                                                                             \
  const float cxj = dequantize_half((b).x & 0xFFFFu), cyj = dequantize_half((b).x >> 16), czj = dequantize_half((b).y & 0xFFFFu); \
  const float exj = dequantize_half((b).z & 0xFFFFu), eyj = dequantize_half((b).z >> 16), ezj = dequantize_half((b).w & 0xFFFFu)
    uint64_t okb[G];  // lanes that exist and pass the cone test (per view when the views have different camera positions)
    auto cone_pass = [&](const kconst32p row, float camx, float camy, float camz) {
      ConeU cu;
#pragma unroll
      for (int k = 0; k < 9; k++) cu.nm[k] = asf(row[kRowNm + k]);
#pragma unroll
      for (int k = 0; k < 6; k++) cu.w2[k >> 1][k & 1] = asf(row[kRowWorld2 + k]);
#pragma unroll
      for (int k = 0; k < 2; k++) cu.wt2[k] = asf(row[kRowWorldT2 + k]);
#pragma unroll
      for (int k = 0; k < 4; k++) cu.wr2[k] = asf(row[kRowWorldR2 + k]);
      cu.scale_max = asf(row[kRowScale]);
#pragma unroll
      for (int j = 0; j < G; j++) {
        const uint4 b = bnd[j];
        const bool valid = k0 + (uint32_t)j * 64u + (uint32_t)lane < count;
        const bool nc = valid && ((int32_t)b.w >> 24) != 127;  // cutoff >= 1.0 <=> s8 == 127: cone test skipped
        bool ok = valid;
        if (__builtin_amdgcn_ballot_w64(nc)) {  // wave-uniform
          OXC_MV_DECODE(b);
          const f2 axy = s8_over_127_x2((int32_t)(b.y << 8) >> 24, (int32_t)b.y >> 24);
          const f2 azc = s8_over_127_x2((int32_t)(b.w << 8) >> 24, (int32_t)b.w >> 24);
          const int tier1 = cone_visible_fast(cu, camx, camy, camz, cxj, cyj, czj, exj, eyj, ezj, axy.x, axy.y, azc.x, azc.y);
          bool cone_ok = tier1 == 1;
          if (__builtin_amdgcn_ballot_w64(nc && tier1 == 2)) {  // some lane sits within the margin: the canonical IEEE path decides
            const bool exact = cone_visible(cu, camx, camy, camz, cxj, cyj, czj, exj, eyj, ezj, axy.x, axy.y, azc.x, azc.y);
            cone_ok = tier1 == 2 ? exact : cone_ok;
          }
          ok = nc ? cone_ok : ok;
        }
        okb[j] = __builtin_amdgcn_ballot_w64(ok);
      }
    };
    const MvView* w0 = a.dev + (uint32_t)__builtin_ctz(vmask);  // the group's leader
    const kconst32p row0 = const_row(w0->rows, mi);
    if (SAME_POS) cone_pass(row0, w0->cam_pos[0], w0->cam_pos[1], w0->cam_pos[2]);
    // The six plane distances of every box against the leader's plane normals, once (frustum_plane_dots: the expressions of
    // test_frustum_planes).  A view with the same normals -- nmask -- only compares them with its own offsets.
    f2 dd[G][3];
    {
      float rpl[24], rsg[18];
#pragma unroll
      for (int k = 0; k < 24; k++) rpl[k] = asf(row0[kRowPlanes + k]);
#pragma unroll
      for (int k = 0; k < 18; k++) rsg[k] = asf(row0[kRowSigns + k]);
#pragma unroll
      for (int j = 0; j < G; j++) {
        OXC_MV_DECODE(bnd[j]);
        frustum_plane_dots(rpl, rsg, cxj, cyj, czj, exj, eyj, ezj, dd[j]);
      }
    }
    uint32_t blo = 0, bhi = 0;  // lane v * G + j: ballot j of view v
    uint32_t cnts = 0;          // lane v: survivors of view v in this chunk
    for (uint32_t m = vmask; m; m &= m - 1u) {
      const uint32_t v = (uint32_t)__builtin_ctz(m);
      uint64_t bits[G];
      if (SAME_POS && ((nmask >> v) & 1u)) {  // (wave-uniform)
        float nw[6];
#pragma unroll
        for (int k = 0; k < 6; k++) nw[k] = asf(readlane_u(asu(lw[k]), (int)v));
#pragma unroll
        for (int j = 0; j < G; j++) {
          bool in = true;
#pragma unroll
          for (int p = 0; p < 3; p++) in = in & !(dd[j][p].x <= nw[2 * p]) & !(dd[j][p].y <= nw[2 * p + 1]);
          bits[j] = __builtin_amdgcn_ballot_w64(in) & okb[j];
        }
      } else {
        const MvView* w = a.dev + v;
        const kconst32p row = const_row(w->rows, mi);
        if (!SAME_POS) cone_pass(row, w->cam_pos[0], w->cam_pos[1], w->cam_pos[2]);
        float pl[24], sg[18];
#pragma unroll
        for (int k = 0; k < 24; k++) pl[k] = asf(row[kRowPlanes + k]);
#pragma unroll
        for (int k = 0; k < 18; k++) sg[k] = asf(row[kRowSigns + k]);
#pragma unroll
        for (int j = 0; j < G; j++) {
          // (SAME_POS: the records are fetched again here -- L2 hits, a rare path -- instead of being held across the view loop for its sake:
          // 16 VGPRs that decide between 4 and 5 waves per SIMD)
          const uint4 bj = SAME_POS ? load_global_u4(bounds, min(k0 + (uint32_t)j * 64u + (uint32_t)lane, count - 1u)) : bnd[j];
          OXC_MV_DECODE(bj);
          bits[j] = __builtin_amdgcn_ballot_w64(test_frustum_planes(pl, sg, cxj, cyj, czj, exj, eyj, ezj)) & okb[j];
        }
      }
      uint32_t cnt = 0;
#pragma unroll
      for (int j = 0; j < G; j++) {
        cnt += (uint32_t)__popcll((unsigned long long)bits[j]);
        const bool here = (uint32_t)lane == v * (uint32_t)G + (uint32_t)j;  // (a select, not v_writelane: its lane index would have to travel in m0)
        blo = here ? (uint32_t)bits[j] : blo;
        bhi = here ? (uint32_t)(bits[j] >> 32) : bhi;
      }
      cnts = (uint32_t)lane == v ? cnt : cnts;
    }
#undef OXC_MV_DECODE
    // ---- results: lane v * G + j stores ballot j of view v (its view's 32-byte run), lane v the view's count / id base
    {
      static_assert(G == 4 && kMaxBatch == 16, "one ballot word per lane");
      const uint32_t vv = (uint32_t)lane >> 2, jj = (uint32_t)lane & 3u;
      const uint32_t pv = (uint32_t)__shfl((int)lp, (int)vv, 64);
      if ((vmask >> vv) & 1u) gptr(a.dev[vv].bits)[(size_t)pv * G + jj] = (uint64_t)blo | ((uint64_t)bhi << 32);
      if ((uint32_t)lane < 16u && ((vmask >> lane) & 1u)) {
        const MvView* w = a.dev + lane;
        gptr(w->counts)[lp] = cnts;
        gptr(w->idbase)[lp] = lid0;
        if (cnts) __hip_atomic_fetch_add(gptr(w->supers) + (lp / kChunksPerSuper) * kSuperStride, cnts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    step = OXC_TICKET_STEP(readlane_u(next_ticket, 0), K, kx);
  }
}

// (5) per view: ordered expansion of the view's chunk ballots into its visible list.  One WAVE per span of 16 chunks (64 ballots): its
// loads (ballots, id bases, the counts in front of it) are all issued at once and nothing waits for a block barrier -- the
// block-per-span form of the single-view emit kernel spent its time in two dependent round trips per span (74 -> us per 16 views).
__global__ __launch_bounds__(256) void k_mv_emit(MvArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const MvView& w = a.dev[blockIdx.y];
  // (the view's pointers as values: read through the reference, `w.out` was re-loaded from the view table -- a scalar load and its wait -- in front of every
  //  store of the expansion loop below, since the compiler cannot know that the stores leave the table alone; round 6)
  uint32_t* const out = w.out;
  const uint64_t* const wbits = w.bits;
  const uint32_t *const widbase = w.idbase, *const wsupers = w.supers, *const wcounts = w.counts;
  uint32_t* const wtri_cmd = w.tri_cmd;
  const uint32_t nchunks = gptr(w.scan_total)[0];
  const uint32_t nwords = nchunks * 4u;
  const uint32_t nspans = (nchunks + 15u) / 16u;
  if (nspans == 0u && blockIdx.x == 0 && threadIdx.x == 0) gptr(wtri_cmd)[0] = 0u;
  for (uint32_t span = blockIdx.x * 4u + (uint32_t)wave; span < nspans; span += gridDim.x * 4u) {
    const uint32_t wd = span * 64u + (uint32_t)lane;
    const uint32_t wc = min(wd, nwords - 1u);
    uint64_t bits = gptr(wbits)[wc];
    const uint32_t idb = gptr(widbase)[wc >> 2] + (wc & 3u) * 64u;
    // exclusive base of the span: all supers before its super + the chunk counts inside it
    const uint32_t c0 = span * 16u, sup = c0 / kChunksPerSuper;
    uint32_t acc = 0;
    // (eight loads in flight per lane: a view of 10 M meshlets has 610 super-chunks; measured on configs[4]: no change, the kernel streams its
    // 156 MB of ids at the 3.5 TB/s default-policy stores reach)
    for (uint32_t i0 = (uint32_t)lane; i0 < sup; i0 += 512u) {
      uint32_t part[8];
#pragma unroll
      for (uint32_t u = 0; u < 8u; u++) part[u] = gptr(wsupers)[min(i0 + 64u * u, sup - 1u) * kSuperStride];
#pragma unroll
      for (uint32_t u = 0; u < 8u; u++) acc += i0 + 64u * u < sup ? part[u] : 0u;
    }
    const uint32_t j = sup * kChunksPerSuper + (uint32_t)lane;
    if (j < c0) acc += gptr(wcounts)[j];
    const uint32_t base = wave_sum(acc);
    if (wd >= nwords) bits = 0ull;
    const uint32_t cnt = (uint32_t)__popcll((unsigned long long)bits);
    const uint32_t incl = wave_incl_scan(cnt, lane);
    const uint32_t off = base + incl - cnt;
    if (lane == 63 && span == nspans - 1u) gptr(wtri_cmd)[0] = base + incl;  // cull_triangles_cmd.x
    const uint32_t blo = (uint32_t)bits, bhi = (uint32_t)(bits >> 32);
    // (Staging the span's ids in an LDS run and writing 16-byte stores, as expand_slots_wide does for the index list: 44 -> 115 us with either store
    // policy -- the second pass over the 64 words and the LDS hand-offs cost more than the 64 short stores.)
#pragma unroll 4
    for (int k = 0; k < 64; k++) {
      const uint64_t bk = (uint64_t)readlane_u(blo, k) | ((uint64_t)readlane_u(bhi, k) << 32);
      if (bk == 0ull) continue;  // wave-uniform
      const uint32_t ok = readlane_u(off, k), ik = readlane_u(idb, k);
      // (a lane's rank among the word's set bits: v_mbcnt with the word as its scalar mask operand)
      if ((bk >> lane) & 1ull) gptr(out)[ok + __builtin_amdgcn_mbcnt_hi((uint32_t)(bk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bk, 0u))] = ik + (uint32_t)lane;
    }
  }
}

void launch_mv_setup(const MvArgs& a, const MvBlob& blob, uint32_t grid, hipStream_t s) {
  hipLaunchKernelGGL(k_mv_group, dim3(grid), dim3(256), 0, s, a, blob);
  hipLaunchKernelGGL(k_mv_scan, dim3(1, a.views + 1), dim3(1024), 0, s, a);
  hipLaunchKernelGGL(k_mv_steps, dim3(std::max(1u, std::min((a.M + 3u) / 4u, grid))), dim3(256), 0, s, a);
}
void launch_mv_test(const MvArgs& a, uint32_t num_cus, hipStream_t s) {
  if (a.same_pos)
    hipLaunchKernelGGL(k_mv_test<true>, dim3(num_cus * 8u), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(k_mv_test<false>, dim3(num_cus * 8u), dim3(256), 0, s, a);
}
void launch_mv_emit(const MvArgs& a, uint32_t max_chunks_per_view, uint32_t max_grid, hipStream_t s) {
  const uint32_t blocks = std::max(1u, ((max_chunks_per_view + 15u) / 16u + 3u) / 4u);  // one wave per span
  hipLaunchKernelGGL(k_mv_emit, dim3(std::min(blocks, std::max(max_grid / std::max(1u, a.views), max_grid / 4u)), a.views), dim3(256), 0, s, a);
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
void launch_prepare(const PrepareArgs& a, uint32_t grid, uint32_t views, hipStream_t s) {
  hipLaunchKernelGGL(k_prepare_instances, dim3(grid, 1 + views), dim3(256), 0, s, a);
}

void launch_scan_mesh_counts(const uint32_t* counts, uint32_t* offsets, uint32_t n, uint32_t cap, uint32_t* vis, uint32_t* cmd, hipStream_t s) {
  ScanArgs a{counts, offsets, n, cap, vis, cmd};
  hipLaunchKernelGGL(k_scan_mesh_counts, dim3(1), dim3(1024), 0, s, a);
}
void launch_expand(const uint32_t* counts, const uint32_t* offsets, uint32_t n, uint32_t cap, void* out, uint32_t grid, hipStream_t s) {
  ExpandArgs a{counts, offsets, n, cap, reinterpret_cast<GpuMeshletInstance*>(out)};
  hipLaunchKernelGGL(k_expand_meshlet_instances, dim3(grid), dim3(256), 0, s, a);
}
void launch_prepare_batch(const BatchBlob& blob, BatchElem* dev, uint32_t grid, hipStream_t s) {
  hipLaunchKernelGGL(k_prepare_batch, dim3(grid, blob.count), dim3(256), 0, s, blob, dev);
}
void launch_scan_batch(const BatchElem* dev, uint32_t count, hipStream_t s) { hipLaunchKernelGGL(k_scan_batch, dim3(1, count), dim3(1024), 0, s, dev); }
void launch_expand_batch(const BatchElem* dev, uint32_t count, uint32_t grid, hipStream_t s) {
  hipLaunchKernelGGL(k_expand_batch, dim3(grid, count), dim3(256), 0, s, dev);
}
void launch_meshlets_test_batch(const BatchElem* dev, uint32_t count, uint32_t grid, hipStream_t s) {
  hipLaunchKernelGGL(k_cull_meshlets_test_batch, dim3(grid * (4 / kPlainBlockWaves), count), dim3(64 * kPlainBlockWaves), 0, s, dev);
}
void launch_meshlets_emit_batch(const BatchElem* dev, uint32_t count, uint32_t grid, hipStream_t s) {
  hipLaunchKernelGGL(k_cull_meshlets_emit_batch, dim3(grid, count), dim3(256), 0, s, dev);
}
void launch_tris_test_batch(const BatchElem* dev, uint32_t count, uint32_t grid, hipStream_t s) {
  hipLaunchKernelGGL(k_cull_triangles_test_batch, dim3(grid, count), dim3(256), 0, s, dev);
}
void launch_tris_emit_batch(const BatchElem* dev, uint32_t count, uint32_t grid, hipStream_t s) {
  hipLaunchKernelGGL(k_cull_triangles_emit_batch, dim3(grid, count), dim3(256), 0, s, dev);
}

constexpr int kHizGroups = (int)kHizGroupsPerWave;  // groups per wave of the HiZ variants (block = 16 / kHizGroups waves)
// The HiZ variants stage ~22 KB of the pyramid per block and loop over chunks: a grid of exactly the resident block
// count (occupancy query) runs one round of blocks; with the generic 8 blocks per CU the second half of the grid starts
// late and stages the pyramid again (measured on config 3: 112 -> 106 us per launch; 1.25x / 1.5x the resident count
// are worse than both).
template <class K>
static uint32_t resident_grid(K kernel, uint32_t block, uint32_t num_cus) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, (int)block, 0) != hipSuccess || per_cu <= 0) return num_cus * 8u;
  return (uint32_t)per_cu * num_cus;
}
void launch_meshlets_test(const MeshletTestArgs& a, bool hiz, bool occl, bool late, uint32_t grid, uint32_t num_cus, uint32_t grid_limit, hipStream_t s) {
  constexpr uint32_t hb = 1024 / kHizGroups;
  if (hiz && grid_limit) grid = std::min(grid, grid_limit);
  if (a.out) {  // unordered_output: the appending instantiation of the plain kernel (the HiZ kernels keep the ordered form)
    hipLaunchKernelGGL(k_cull_meshlets_test_unordered, dim3((grid * 4 + kUnordBlockWaves - 1) / kUnordBlockWaves), dim3(64 * kUnordBlockWaves), 0, s, a);
    return;
  }
  if (hiz && occl && a.dbg_occlusion) {  // (measurement aid: not a timed path)
    const dim3 g(std::min(grid, num_cus * 4u)), b(hb);
    if (a.share == 2)
      hipLaunchKernelGGL((k_cull_meshlets_test_counting<true, 2>), g, b, 0, s, a);
    else if (a.share == 1)
      hipLaunchKernelGGL((k_cull_meshlets_test_counting<false, 1>), g, b, 0, s, a);
    else if (late)
      hipLaunchKernelGGL((k_cull_meshlets_test_counting<true, 0>), g, b, 0, s, a);
    else
      hipLaunchKernelGGL((k_cull_meshlets_test_counting<false, 0>), g, b, 0, s, a);
    return;
  }
  if (!hiz) {
    hipLaunchKernelGGL((k_cull_meshlets_test<false, false, false>), dim3(grid * (4 / kPlainBlockWaves)), dim3(64 * kPlainBlockWaves), 0, s, a);
  } else if (occl && a.share) {
    if (a.share == 2) {
      static const uint32_t cap = resident_grid(k_cull_meshlets_test_shared<true>, hb, num_cus);
      hipLaunchKernelGGL((k_cull_meshlets_test_shared<true>), dim3(std::min(grid, cap)), dim3(hb), 0, s, a);
    } else {
      static const uint32_t cap = resident_grid(k_cull_meshlets_test_shared<false>, hb, num_cus);
      hipLaunchKernelGGL((k_cull_meshlets_test_shared<false>), dim3(std::min(grid, cap)), dim3(hb), 0, s, a);
    }
  } else if (occl && late) {
    static const uint32_t cap = resident_grid(k_cull_meshlets_test<true, true, true, kHizGroups>, hb, num_cus);
    hipLaunchKernelGGL((k_cull_meshlets_test<true, true, true, kHizGroups>), dim3(std::min(grid, cap)), dim3(hb), 0, s, a);
  } else if (occl) {
    static const uint32_t cap = resident_grid(k_cull_meshlets_test<true, true, false, kHizGroups>, hb, num_cus);
    hipLaunchKernelGGL((k_cull_meshlets_test<true, true, false, kHizGroups>), dim3(std::min(grid, cap)), dim3(hb), 0, s, a);
  } else if (late) {
    static const uint32_t cap = resident_grid(k_cull_meshlets_test<true, false, true, kHizGroups>, hb, num_cus);
    hipLaunchKernelGGL((k_cull_meshlets_test<true, false, true, kHizGroups>), dim3(std::min(grid, cap)), dim3(hb), 0, s, a);
  } else {
    static const uint32_t cap = resident_grid(k_cull_meshlets_test<true, false, false, kHizGroups>, hb, num_cus);
    hipLaunchKernelGGL((k_cull_meshlets_test<true, false, false, kHizGroups>), dim3(std::min(grid, cap)), dim3(hb), 0, s, a);
  }
}
void launch_hpb_test(const HpbTestArgs& a, uint32_t grid, hipStream_t s) { hipLaunchKernelGGL(k_cull_meshlets_hpb_test, dim3(grid), dim3(256), 0, s, a); }
void launch_meshlets_emit(const MeshletEmitArgs& a, bool hiz, bool late, uint32_t grid, hipStream_t s) {
  dim3 g(grid), b(256);
  if (!hiz)
    hipLaunchKernelGGL((k_cull_meshlets_emit<false, false>), g, b, 0, s, a);
  else if (late)
    hipLaunchKernelGGL((k_cull_meshlets_emit<true, true>), g, b, 0, s, a);
  else
    hipLaunchKernelGGL((k_cull_meshlets_emit<true, false>), g, b, 0, s, a);
}
void launch_tris_test(const TriTestArgs& a, bool late, bool wide, bool small_triangle_cull, bool cached, uint32_t grid, hipStream_t s) {
  dim3 g(grid), b(256);
  if (cached && !small_triangle_cull) {  // shared geometry (the small-triangle instantiations keep the streaming loads)
    if (late && wide)
      hipLaunchKernelGGL((k_cull_triangles_test_cached<true, true>), g, b, 0, s, a);
    else if (late)
      hipLaunchKernelGGL((k_cull_triangles_test_cached<true, false>), g, b, 0, s, a);
    else if (wide)
      hipLaunchKernelGGL((k_cull_triangles_test_cached<false, true>), g, b, 0, s, a);
    else
      hipLaunchKernelGGL((k_cull_triangles_test_cached<false, false>), g, b, 0, s, a);
    return;
  }
  const int v = (late ? 4 : 0) | (wide ? 2 : 0) | (small_triangle_cull ? 1 : 0);
  switch (v) {
    case 0: hipLaunchKernelGGL((k_cull_triangles_test<false, false, false>), g, b, 0, s, a); break;
    case 1: hipLaunchKernelGGL((k_cull_triangles_test<false, false, true>), g, b, 0, s, a); break;
    case 2: hipLaunchKernelGGL((k_cull_triangles_test<false, true, false>), g, b, 0, s, a); break;
    case 3: hipLaunchKernelGGL((k_cull_triangles_test<false, true, true>), g, b, 0, s, a); break;
    case 4: hipLaunchKernelGGL((k_cull_triangles_test<true, false, false>), g, b, 0, s, a); break;
    case 5: hipLaunchKernelGGL((k_cull_triangles_test<true, false, true>), g, b, 0, s, a); break;
    case 6: hipLaunchKernelGGL((k_cull_triangles_test<true, true, false>), g, b, 0, s, a); break;
    default: hipLaunchKernelGGL((k_cull_triangles_test<true, true, true>), g, b, 0, s, a); break;
  }
}
// resident_cus != 0: the grid is also capped at the blocks of THIS instantiation that are resident at once on that many CUs (occupancy
// query).  The fused kernel hands out its work by rounds of the grid (tris_fused_body): a grid of two resident rounds -- the WIDE
// instantiations run 4 waves per SIMD, the generic cap was 8 blocks per CU -- cost the 8 M x 124-triangle frame 33 us (0.562 -> 0.529 ms).
void launch_tris_fused(const TriTestArgs& a, bool late, uint32_t wide, bool small_triangle_cull, bool cached, uint32_t grid, uint32_t resident_cus, hipStream_t s) {
  dim3 b(256);
  const int v = (late ? 4 : 0) | (wide ? 2 : 0) | (small_triangle_cull ? 1 : 0);
#define OXC_FUSED_LAUNCH(K_)                                                                                       \
  {                                                                                                                \
    static const uint32_t per_cu = resident_grid(K_, 256, 1);                                                      \
    const uint32_t g = resident_cus ? std::min(grid, per_cu * resident_cus) : grid;                                \
    hipLaunchKernelGGL(K_, dim3(std::max(g, 1u)), b, 0, s, a);                                                     \
  }
#define OXC_FUSED_CASE(i, L_, W_, S_)                                                                              \
  case i:                                                                                                          \
    OXC_FUSED_LAUNCH((k_cull_triangles_fused<L_, W_, S_>))                                                         \
    break;
  if (cached && !small_triangle_cull) {  // shared geometry: plain loads
    switch ((late ? 4 : 0) | (int)std::min(wide, 2u)) {
      case 0: OXC_FUSED_LAUNCH((k_cull_triangles_fused_cached<false, 0>)) break;
      case 1: OXC_FUSED_LAUNCH((k_cull_triangles_fused_cached<false, 1>)) break;
      case 2: OXC_FUSED_LAUNCH((k_cull_triangles_fused_cached<false, 2>)) break;
      case 4: OXC_FUSED_LAUNCH((k_cull_triangles_fused_cached<true, 0>)) break;
      case 5: OXC_FUSED_LAUNCH((k_cull_triangles_fused_cached<true, 1>)) break;
      default: OXC_FUSED_LAUNCH((k_cull_triangles_fused_cached<true, 2>)) break;
    }
    return;
  }
  if (wide == 2u) {  // {id, corner} pairs
    switch (v & 5) {
      case 0: OXC_FUSED_LAUNCH((k_cull_triangles_fused_pairs<false, false>)) break;
      case 1: OXC_FUSED_LAUNCH((k_cull_triangles_fused_pairs<false, true>)) break;
      case 4: OXC_FUSED_LAUNCH((k_cull_triangles_fused_pairs<true, false>)) break;
      default: OXC_FUSED_LAUNCH((k_cull_triangles_fused_pairs<true, true>)) break;
    }
    return;
  }
  switch (v) {
    OXC_FUSED_CASE(0, false, false, false)
    OXC_FUSED_CASE(1, false, false, true)
    OXC_FUSED_CASE(2, false, true, false)
    OXC_FUSED_CASE(3, false, true, true)
    OXC_FUSED_CASE(4, true, false, false)
    OXC_FUSED_CASE(5, true, false, true)
    OXC_FUSED_CASE(6, true, true, false)
    default:
      OXC_FUSED_CASE(7, true, true, true)
  }
#undef OXC_FUSED_CASE
#undef OXC_FUSED_LAUNCH
}
void launch_tris_emit(const TriEmitArgs& a, bool late, uint32_t wide, uint32_t grid, hipStream_t s) {
  dim3 g(grid), b(256);
  if (wide == 2u && late)
    hipLaunchKernelGGL((k_cull_triangles_emit_pairs<true>), g, b, 0, s, a);
  else if (wide == 2u)
    hipLaunchKernelGGL((k_cull_triangles_emit_pairs<false>), g, b, 0, s, a);
  else if (late && wide)
    hipLaunchKernelGGL((k_cull_triangles_emit<true, true>), g, b, 0, s, a);
  else if (late)
    hipLaunchKernelGGL((k_cull_triangles_emit<true, false>), g, b, 0, s, a);
  else if (wide)
    hipLaunchKernelGGL((k_cull_triangles_emit<false, true>), g, b, 0, s, a);
  else
    hipLaunchKernelGGL((k_cull_triangles_emit<false, false>), g, b, 0, s, a);
}
void launch_hiz(const HizArgs& a, uint32_t num_cus, hipStream_t s) {
  if (a.w % 64 == 0 && a.h % 64 == 0) {
    // (A persistent form -- blocks walking tiles with the next tile's loads issued before the current tile's stores, one barrier
    // per tile -- was measured in round 2: 81 us against 77 us, byte-identical.  tools/hiz_probe.hip shows why nothing of that kind
    // helps: reading every other row of an 8192^2 image that is not cache-resident runs at ~4.5 TB/s of line traffic (30 us) in
    // EVERY tile shape, whole rows included; the rest of the kernel is its 90 MB of stores.)
    (void)num_cus;
    hipLaunchKernelGGL(k_hiz_tile, dim3(a.w / 64, a.h / 64), dim3(256), 0, s, a);
    if (a.levels > 7) hipLaunchKernelGGL(k_hiz_tail, dim3(1), dim3(1024), 0, s, a, 6u);
  } else {
    uint32_t n = a.w * a.h;
    hipLaunchKernelGGL(k_hiz_mip0_generic, dim3((n + 255) / 256), dim3(256), 0, s, a);
    if (a.levels > 1) hipLaunchKernelGGL(k_hiz_tail, dim3(1), dim3(1024), 0, s, a, 0u);
  }
}
void launch_seed_slot(uint32_t* slot, uint32_t total, hipStream_t s) { hipLaunchKernelGGL(k_seed_slot, dim3(1), dim3(64), 0, s, slot, total); }
void launch_pack_counters(const uint32_t* vis, const uint32_t* tri_cmd, const uint32_t* draw_cmd, uint32_t* out4, hipStream_t s) {
  hipLaunchKernelGGL(k_pack_counters, dim3(1), dim3(64), 0, s, vis, tri_cmd, draw_cmd, out4);
}
__global__ __launch_bounds__(64) void k_pack_counters_batch(PackBlob b, uint32_t* out4) {
  const uint32_t e = threadIdx.x;
  if (e < b.count) {
    const uint32_t *vis = b.vis[e], *tri = b.tri_cmd[e], *draw = b.draw_cmd[e];
    out4[e * 4 + 0] = tri ? tri[0] : 0u;
    out4[e * 4 + 1] = vis ? vis[0] : 0u;  // (a batched element is a plain call: visibility.total = the length of its MeshletInstance list)
    out4[e * 4 + 2] = vis ? vis[2] : 0u;
    out4[e * 4 + 3] = draw ? draw[0] : 0u;
  }
}
void launch_pack_counters_batch(const PackBlob& blob, uint32_t* out4, hipStream_t s) { hipLaunchKernelGGL(k_pack_counters_batch, dim3(1), dim3(64), 0, s, blob, out4); }
void launch_stream_read(const void* p, uint64_t bytes, uint32_t* sink, uint32_t grid, hipStream_t s) {
  hipLaunchKernelGGL(k_stream_read, dim3(grid), dim3(256), 0, s, reinterpret_cast<const uint4*>(p), bytes / 16, sink);
}
__global__ __launch_bounds__(256) void k_debug_project_aabb(DebugProjectArgs a) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += gridDim.x * blockDim.x) {
    const float* b = a.boxes6 + (size_t)i * 6;
    float sa[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool ok = project_aabb<false>(a.mvp, a.near_clip, b[0], b[1], b[2], b[3], b[4], b[5], sa);
    // ... and the form with the orthographic short cut, which has to agree (the sign of a zero aside): o[6] = 2 marks a box where it does not
    float sb[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool ok_b = project_aabb<true>(a.mvp, a.near_clip, b[0], b[1], b[2], b[3], b[4], b[5], sb);
    bool same = ok == ok_b;
#pragma unroll
    for (int k = 0; k < 6; k++) same = same && (!ok || asu(sa[k]) == asu(sb[k]) || (sa[k] == 0.0f && sb[k] == 0.0f));
    float* o = a.out7 + (size_t)i * 7;
#pragma unroll
    for (int k = 0; k < 6; k++) o[k] = ok ? sa[k] : 0.0f;
    o[6] = !same ? 2.0f : ok ? 1.0f : 0.0f;
  }
}
void launch_debug_project_aabb(const DebugProjectArgs& a, hipStream_t s) {
  if (a.n) hipLaunchKernelGGL(k_debug_project_aabb, dim3((a.n + 255) / 256), dim3(256), 0, s, a);
}
void launch_debug_decode_bounds(const void* bounds, uint32_t n, float* out10, hipStream_t s) {
  hipLaunchKernelGGL(k_debug_decode_bounds, dim3((n + 255) / 256), dim3(256), 0, s, reinterpret_cast<const uint4*>(bounds), n, out10);
}

}  // namespace oxc
