// oxcull_types.hpp -- device-side mirrors of the reference GPU structs and the per-instance
// cache the kernels stream against.  Layouts: Oxylus/include/Scene/SceneGPU.hpp:84-152.
#pragma once
#include <cstddef>
#include <cstdint>

namespace oxc {

struct GpuMeshletInstance {  // SceneGPU.hpp:105-108
  uint32_t mesh_instance_index, meshlet_index;
};
struct GpuMeshInstance {  // SceneGPU.hpp:110-116
  uint32_t mesh_index, lod_index, material_index, transform_index, meshlet_instance_visibility_offset;
};
struct GpuMeshlet {  // SceneGPU.hpp:118-123
  uint32_t indirect_vertex_index_offset, local_triangle_index_offset, vertex_count, triangle_count;
};
struct GpuMeshLOD {  // SceneGPU.hpp:125-139
  uint64_t indices, meshlets, meshlet_bounds, local_triangle_indices, indirect_vertex_indices;
  uint32_t indices_count, meshlet_count, meshlet_bounds_count, local_triangle_indices_count,
      indirect_vertex_indices_count;
  float error;
};
struct GpuMesh {  // SceneGPU.hpp:141-152
  uint64_t vertex_positions, vertex_normals, texture_coords;
  uint32_t vertex_count, lod_count;
  uint64_t lods;
  float aabb_center[3];
  float aabb_extent[3];
};
static_assert(sizeof(GpuMeshletInstance) == 8, "layout");
static_assert(sizeof(GpuMeshInstance) == 20, "layout");
static_assert(sizeof(GpuMeshlet) == 16, "layout");
static_assert(sizeof(GpuMeshLOD) == 64, "layout");
static_assert(sizeof(GpuMesh) == 64, "layout");

// Everything a meshlet/triangle test needs about its mesh instance, resolved once per call by
// k_prepare_instances instead of once per meshlet (the reference re-derives mvp/planes/normal
// matrix in every thread and chases mesh_instance -> mesh -> lods[lod] -> pointer per meshlet,
// cull_meshlets.slang:37-52).  Same arithmetic, same order => same bits.
// The row is read by scalar loads (s_load_dwordx16 and friends).  Wave-uniform operands are laid
// out in PAIRS so that, once in SGPRs, they feed the packed f32 VALU ops (v_pk_mul_f32 /
// v_pk_add_f32: two IEEE binary32 operations per instruction, no contraction) directly.
struct alignas(64) InstCache {
  // dwords 0..23: the 6 normalised frustum planes of mvp (cull.slang:58-71), two planes side by side:
  // planes2[p][c][k] = component c (x,y,z,w) of plane 2p+k
  float planes2[3][4][2];
  // dwords 24..41: signs2[p][c][k] = -1.0 if the sign bit of planes2[p][c][k] is set, else +1.0 (c = x,y,z)
  float signs2[3][3][2];
  uint32_t vis_offset;     // 42: MeshInstance::meshlet_instance_visibility_offset
  uint32_t meshlet_count;  // 43: of the selected LOD
  float mvp[16];           // 44..59: projection_view * world, column-major
  float scale_max;         // 60: max row length of world's 3x3 (scene.slang:305-310)
  uint32_t _pad0[3];
  // dwords 64..95 (second load)
  float nm[9];             // 64..72: TransformWorld::normal_matrix(), column-major (scene.slang:292-299)
  uint32_t transform_index;  // 73: MeshInstance::transform_index (the multi-view meshlet test groups the views that share it)
  float world2[3][2];      // 74..79: world2[c][k] = world(row k, col c), rows 0 and 1 side by side
  float world_t2[2];       // 80..81: world(row 0, col 3), world(row 1, col 3)
  float world_r2[4];       // 82..85: row 2 of world
  uint64_t bounds;     // 86: MeshLOD::meshlet_bounds
  uint64_t meshlets;   // MeshLOD::meshlets
  uint64_t micro;      // MeshLOD::local_triangle_indices
  uint64_t vidx;       // MeshLOD::indirect_vertex_indices
  uint64_t positions;  // Mesh::vertex_positions
};
static_assert(sizeof(InstCache) == 384, "layout");
// dword offsets of the fields: what the kernels' scalar loads index
enum : int {
  kRowPlanes = 0, kRowSigns = 24, kRowVisOffset = 42, kRowMeshletCount = 43, kRowMvp = 44, kRowScale = 60,
  kRowNm = 64, kRowTransformIndex = 73, kRowWorld2 = 74, kRowWorldT2 = 80, kRowWorldR2 = 82, kRowBounds = 86,
};
static_assert(offsetof(InstCache, signs2) == kRowSigns * 4 && offsetof(InstCache, vis_offset) == kRowVisOffset * 4 &&
              offsetof(InstCache, mvp) == kRowMvp * 4 && offsetof(InstCache, scale_max) == kRowScale * 4 &&
              offsetof(InstCache, nm) == kRowNm * 4 && offsetof(InstCache, world2) == kRowWorld2 * 4 &&
              offsetof(InstCache, world_t2) == kRowWorldT2 * 4 && offsetof(InstCache, world_r2) == kRowWorldR2 * 4 &&
              offsetof(InstCache, bounds) == kRowBounds * 4, "layout");

// Counter slot handed out per cull_geometry call (u32 indices).
enum : uint32_t {
  SLOT_VIS = 0,         // {total, early, late}
  SLOT_MESHLETS_CMD = 4,  // {x,1,1}
  SLOT_TRI_CMD = 8,     // {x,1,1}
  SLOT_DRAW_CMD = 12,   // {indexCount, instanceCount, firstIndex, vertexOffset, firstInstance}
  SLOT_U32S = 32        // 128 B per slot
};

// Work decomposition constants (see DESIGN.md "ordered compaction").
constexpr uint32_t kGroupsPerWave = 4;       // 64-meshlet groups each wave keeps in flight per block iteration
constexpr uint32_t kPlainGroups = 4;  // groups per wave of the plain (non-HiZ) test kernel; block = 16 / kPlainGroups waves
constexpr uint32_t kHizGroupsPerWave = 4;  // groups per wave of the HiZ test kernels
constexpr uint32_t kPlainBlockWaves = 4;  // the plain test kernel's waves are independent: block size is a scheduling knob
// The appending plain test (unordered_output) spends ONE returning atomic on cull_triangles_cmd.x per block iteration: the block size is how
// many meshlets share an atomic (64 * G per wave).  Measured on configs[1] as one call (1 M meshlets, 977 block atomics at 4 waves):
// 17.9 us per call at 4 waves, 17.5 at 8, 17.8 at 16 -- the counter is not what the call waits for (its two launches' ramp and drain are).
constexpr uint32_t kUnordBlockWaves = 4;
constexpr uint32_t kMeshletChunk = 256 * kGroupsPerWave;  // meshlets per block iteration of the test kernel
constexpr uint32_t kMeshletSpan = 4096;      // meshlets per block iteration of the emit kernel (8 chunks)
constexpr uint32_t kTriChunk = 64;           // visible meshlets per block iteration of the triangle test kernel
constexpr uint32_t kTriSpan = 256;           // visible meshlets per block iteration of the triangle emit kernel
constexpr uint32_t kFusedTriSpan = 128;      // ... and of the fused (unordered_output) kernel: visible meshlets per atomic_add on index_count (configs[2] frame, us: 256 -> 573, 128 -> 562, 64 -> 562)
constexpr uint32_t kHizLdsTexels = 384;      // LDS budget (floats) for the staged top HiZ mips: the 16x16 level and everything above it (round 2: staging the 64x64 level too -- 22 KB per block -- measured 4 % slower: 93 / 121 us against 89 / 116)
constexpr uint32_t kSuperStride = 64;         // words between super-chunk accumulators: one per 256 B so their atomics do not serialise on a cache line
constexpr uint32_t kTicketCounters = 256;    // dynamic work counters of the HiZ meshlet test: counter x hands out the wave steps congruent to x mod 256
constexpr uint32_t kChunksPerSuper = 64;     // chunk counts are also accumulated per 64 chunks
// async_triangles (include/oxcull.h): resident blocks per CU the persistent kernels of the two stages take while they share the machine
// (0 = no limit).  Measured on configs[2] (tools/kbench.py, us per frame; in order on one stream: 613): no limit 623, meshlet / triangle
// blocks per CU 3 / 8: 628, 3 / 5: 639, 4 / 4: 640, 3 / 4: 648, 2 / 5: 685, 2 / 6: 689, 1 / 6: 954 -- the kernels do run side by
// side (their HIP-event times add up to 1.8x the frame) but each slows down by what the other takes: the frame is bound by HBM
// traffic, which both stages draw on (2.4-3.7 TB/s by the meshlet tests, ~5 TB/s by the triangle stage).  Hence no limit by default.
constexpr uint32_t kTriangleBlocksPerCU = 8;  // grid cap of the triangle test / emit / fused kernels, in blocks per CU (oxc_debug_set_tuning moves it for measurements)
constexpr uint32_t kAsyncMeshletBlocksPerCU = 0;
constexpr uint32_t kAsyncTriangleBlocksPerCU = 0;

}  // namespace oxc
