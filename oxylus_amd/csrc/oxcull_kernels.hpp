// oxcull_kernels.hpp -- kernel argument blocks and launcher prototypes shared by the kernel TU
// (oxcull_kernels.hip) and the C-ABI host TU (oxcull_abi.cpp).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "oxcull.h"
#include "oxcull_types.hpp"

namespace oxc {

struct PrepareArgs {
  const GpuMesh* meshes;
  const float* transforms;
  GpuMeshInstance* mesh_instances;
  InstCache* cache;
  uint32_t* mesh_counts;
  uint32_t* slot;          // this call's counter slot
  uint32_t* vis;           // {total, early, late}
  uint32_t* meshlets_cmd;  // {x,1,1}
  uint32_t* supers_meshlets;
  uint32_t* supers_tris;
  uint32_t n_supers_meshlets, n_supers_tris;
  uint32_t* tickets;  // kTicketCounters work counters (stride kSuperStride) zeroed here for the test kernels that take work dynamically; may be null
  // share_pass_tests, early call: the late call of the frame will reuse this call's instance rows, so everything else its prepare
  // kernel would do is done here -- a second set of accumulators and its counter slot -- and the late call launches none.  Null: not armed.
  uint32_t* slot_late;
  uint32_t* supers_meshlets_late;
  uint32_t* supers_tris_late;
  uint32_t* tickets_late;
  uint32_t mesh_instance_count;
  uint32_t cull_flags;
  uint32_t do_cull_meshes;
  uint32_t init_vis;    // write vis/meshlets_cmd initial values
  uint32_t seed_total;  // initial vis.total (0 for the reference flow)
  oxc_cull_camera cam;
  // multi-view (use_hpb): blockIdx.y = 1 + v computes rows view_cache[v * M + mi] with
  // clipmaps[v].projection_view_mat in place of the camera matrix
  const oxc_virtual_clipmap* clipmaps;
  InstCache* view_cache;
};

struct HpbTestArgs {
  const InstCache* cache;
  const InstCache* view_cache;
  const GpuMeshletInstance* meshlet_instances;
  const uint32_t* vis;
  uint64_t* bits;
  uint32_t* chunk_counts;
  uint32_t* supers;
  uint32_t* tickets;  // work counters (zeroed by prepare)
  const oxc_virtual_clipmap* clipmaps;
  const uint32_t* dirty;
  uint32_t clipmap_count;
  uint32_t mesh_instance_count;
  uint32_t n_cap;  // frame.max_meshlet_instance_count: the device-side list length is clamped to what the buffers hold
  const uint8_t* hpb_data;
  uint32_t hpb_w, hpb_h, hpb_layers, hpb_levels;
  uint32_t hpb_level_off[13];
  float light_dir[3];  // camera.position carries -light_dir
};

struct MeshletTestArgs {
  uint32_t n_host;  // != 0: the list length is known on the host (seeded lists); skips the dependent load of vis[0]
  uint32_t n_cap;   // frame.max_meshlet_instance_count: vis[0] is clamped to what the scratch and the output lists hold
  uint32_t mask_bits;  // bits the visibility mask buffer holds; mask indices beyond it read as "not visible" and are never written
  const InstCache* cache;
  const GpuMeshletInstance* meshlet_instances;
  const uint32_t* vis;
  uint32_t* mask;
  uint64_t* bits;
  uint32_t* chunk_counts;
  uint32_t* supers;
  uint32_t* tickets;  // != null: waves take their 64*G-meshlet steps from these counters instead of a fixed stride (zeroed by prepare)
  // Two-pass sharing (oxc_cull_geometry_context::share_pass_tests): the early HiZ call leaves, per 64-meshlet group, the ballot of
  // "passed the frustum and the normal-cone test" (the tests that depend on the camera only) and, per wave step, where its run of mask
  // bits starts; the late call of the same frame reads them instead of testing again.  share: 0 = off, 1 = write them (early call),
  // 2 = read them (late call).
  uint32_t share;
  // unordered_output (include/oxcull.h), plain kernel only: the test kernel appends its survivors itself -- slot allocation by atomic_add as
  // the reference does (cull_meshlets.slang:55-70), aggregated per block through the ballots -- and no emit kernel runs.
  // out = visible_meshlet_instances_indices, count_a = cull_triangles_cmd.x.  Null `out`: the ordered two-launch form.
  uint32_t* out;
  uint32_t* count_a;
  uint32_t* dbg_occlusion;     // measurement aid (null: off): 256 counters, stride kSuperStride words, the counting instantiations add the candidates that reach test_occlusion
  uint64_t* camera_test_bits;  // [ceil(N / 64)]
  uint2* step_info;        // [steps]: {first mask bit of the step, 1 if the step's 64 * G meshlets are one run of mask bits}
  const float* hiz_data;
  uint32_t hiz_level_off[13];  // float offsets of each mip
  uint32_t hiz_w, hiz_h, hiz_levels;
  uint32_t hiz_lds_first;     // first level staged in LDS (== hiz_levels: none)
  uint32_t hiz_lds_off[13];   // float offset of each staged level inside the LDS tile
  float near_clip;
  float cam_pos[3];
};

struct MeshletEmitArgs {
  uint32_t n_host;
  uint32_t n_cap;
  uint32_t count_meshlets;  // meshlets per published count (64 * groups-per-wave of the test kernel that ran)
  const uint64_t* bits;
  const uint32_t* chunk_counts;
  const uint32_t* supers;
  uint32_t* vis;
  uint32_t* tri_cmd;
  uint32_t* out;  // visible_meshlet_instances_indices
};

struct TriTestArgs {
  const InstCache* cache;
  const GpuMeshletInstance* meshlet_instances;
  const uint32_t* visible;
  const uint32_t* vis;
  const uint32_t* tri_cmd;
  uint64_t* tri_masks;
  uint32_t* chunk_counts;
  uint32_t* supers;
  float resolution[2];  // cull_camera.resolution: read by the small-triangle variants only
  // unordered_output: the fused kernel (test + expansion of a kFusedTriSpan-meshlet span in one launch; slots by atomic_add on index_count, as
  // cull_triangles.slang:71-88 does per workgroup) writes the packed indices and the draw command itself
  uint32_t* draw_cmd;
  uint32_t* out;  // reordered_indices
  // round 5, fused kernel: the chunks of the last, partial round of the grid are drawn from these counters (tris_fused_body; zeroed by the
  // prepare kernel: the ordered form's triangle super-chunk accumulators, which the fused form does not use)
  uint32_t* ticket;
  uint32_t ticket_count;  // counters behind `ticket` (stride kSuperStride words), >= 1
};

struct TriEmitArgs {
  const uint64_t* tri_masks;
  const uint32_t* visible;
  const uint32_t* vis;
  const uint32_t* tri_cmd;
  const uint32_t* chunk_counts;
  const uint32_t* supers;
  uint32_t* draw_cmd;
  uint32_t* out;  // reordered_indices
};

struct ScanArgs {
  const uint32_t* counts;
  uint32_t* offsets;
  uint32_t n;
  uint32_t cap;  // frame.max_meshlet_instance_count: the expansion stops there
  uint32_t* vis;
  uint32_t* meshlets_cmd;
};

struct ExpandArgs {
  const uint32_t* counts;
  const uint32_t* offsets;
  uint32_t n;
  uint32_t cap;
  GpuMeshletInstance* out;
};

// Argument blocks of one oxc_cull_geometry_batch call (plain pipeline, <= kMaxBatch independent
// frames).  Passed by value to the batched prepare kernel, which copies it to the context's device
// buffer; every later kernel of the call reads its element through blockIdx.y from there.
constexpr uint32_t kMaxBatch = 16;      // elements per batched call
constexpr uint32_t kBatchPerPrepare = 16;  // element cores (BatchCore) per kernarg segment of k_prepare_batch
struct BatchElem {
  PrepareArgs prep;
  ScanArgs scan;
  ExpandArgs expand;
  MeshletTestArgs test;
  MeshletEmitArgs emit;
  TriTestArgs ttest;
  TriEmitArgs temit;
};
// What the host hands over per element: every pointer and scalar of BatchElem exactly once (the seven stage
// blocks repeat them 2-5 times).  k_prepare_batch rebuilds the stage blocks from it on the device
// (expand_batch_core), so a kernarg segment carries kBatchPerPrepare = 8 elements instead of 5.
struct BatchCore {
  // caller buffers
  const GpuMesh* meshes;
  const float* transforms;
  GpuMeshInstance* mesh_instances;
  GpuMeshletInstance* meshlet_instances;
  uint32_t* visible_out;    // visible_meshlet_instances_indices
  uint32_t* reordered_out;  // reordered_indices
  // the element's scratch lane
  InstCache* cache;
  InstCache* view_cache;
  uint32_t* mesh_counts;
  uint32_t* mesh_offsets;
  uint64_t* bits;
  uint32_t* m_chunk_counts;
  uint32_t* m_supers;
  uint64_t* tri_masks;
  uint32_t* t_chunk_counts;
  uint32_t* t_supers;
  // counters
  uint32_t* slot;  // tri_cmd / draw_cmd live at slot + SLOT_TRI_CMD / SLOT_DRAW_CMD
  uint32_t* vis;
  uint32_t* meshlets_cmd;
  uint32_t n_supers_meshlets, n_supers_tris;
  uint32_t mesh_instance_count;
  uint32_t cull_flags;
  uint32_t do_cull_meshes;
  uint32_t init_vis;
  uint32_t n_host;
  uint32_t n_cap;
  uint32_t count_meshlets;  // meshlets per published count of the test kernel (64 * groups per wave)
  oxc_cull_camera cam;
};
// What one k_prepare_batch launch receives by value: up to kBatchPerPrepare elements, which it expands to
// dev[first .. first + count) and prepares (blockIdx.y = element).
struct BatchBlob {
  BatchCore core[kBatchPerPrepare];
  uint32_t count;
  uint32_t first;
};
static_assert(sizeof(BatchBlob) <= 8000, "must fit the kernarg segment");

// The stage blocks of one element from its core (the only place that knows which block needs which field).
#if defined(__HIPCC__)
__host__ __device__
#endif
inline void prepare_args_of(const BatchCore& c, PrepareArgs& pa) {
  pa.meshes = c.meshes;
  pa.transforms = c.transforms;
  pa.mesh_instances = c.mesh_instances;
  pa.cache = c.cache;
  pa.mesh_counts = c.mesh_counts;
  pa.slot = c.slot;
  pa.vis = c.vis;
  pa.meshlets_cmd = c.meshlets_cmd;
  pa.supers_meshlets = c.m_supers;
  pa.supers_tris = c.t_supers;
  pa.tickets = nullptr;  // batched elements run the plain meshlet test (fixed stride)
  pa.slot_late = nullptr;
  pa.supers_meshlets_late = pa.supers_tris_late = pa.tickets_late = nullptr;
  pa.n_supers_meshlets = c.n_supers_meshlets;
  pa.n_supers_tris = c.n_supers_tris;
  pa.mesh_instance_count = c.mesh_instance_count;
  pa.cull_flags = c.cull_flags;
  pa.do_cull_meshes = c.do_cull_meshes;
  pa.init_vis = c.init_vis;
  pa.seed_total = 0;
  pa.cam = c.cam;
  pa.clipmaps = nullptr;
  pa.view_cache = c.view_cache;
}
#if defined(__HIPCC__)
__host__ __device__
#endif
inline void expand_batch_core(const BatchCore& c, BatchElem& e) {
  uint32_t* tri_cmd = c.slot + SLOT_TRI_CMD;
  uint32_t* draw_cmd = c.slot + SLOT_DRAW_CMD;
  prepare_args_of(c, e.prep);
  e.scan = ScanArgs{c.mesh_counts, c.mesh_offsets, c.mesh_instance_count, c.n_cap, c.vis, c.meshlets_cmd};
  e.expand = ExpandArgs{c.mesh_counts, c.mesh_offsets, c.mesh_instance_count, c.n_cap, c.meshlet_instances};
  MeshletTestArgs& ta = e.test;
  ta.n_host = c.n_host;
  ta.n_cap = c.n_cap;
  ta.mask_bits = 0;
  ta.cache = c.cache;
  ta.meshlet_instances = c.meshlet_instances;
  ta.vis = c.vis;
  ta.mask = nullptr;
  ta.bits = c.bits;
  ta.chunk_counts = c.m_chunk_counts;
  ta.supers = c.m_supers;
  ta.out = ta.count_a = nullptr;  // batched elements keep the ordered form
  ta.dbg_occlusion = nullptr;
  ta.hiz_data = nullptr;
  ta.hiz_w = ta.hiz_h = ta.hiz_levels = ta.hiz_lds_first = 0;
  ta.near_clip = c.cam.near_clip;
  ta.cam_pos[0] = c.cam.position[0];
  ta.cam_pos[1] = c.cam.position[1];
  ta.cam_pos[2] = c.cam.position[2];
  MeshletEmitArgs& ea = e.emit;
  ea.n_host = c.n_host;
  ea.n_cap = c.n_cap;
  ea.count_meshlets = c.count_meshlets;
  ea.bits = c.bits;
  ea.chunk_counts = c.m_chunk_counts;
  ea.supers = c.m_supers;
  ea.vis = c.vis;
  ea.tri_cmd = tri_cmd;
  ea.out = c.visible_out;
  TriTestArgs& tt = e.ttest;
  tt.cache = c.cache;
  tt.meshlet_instances = c.meshlet_instances;
  tt.visible = c.visible_out;
  tt.vis = c.vis;
  tt.tri_cmd = tri_cmd;
  tt.tri_masks = c.tri_masks;
  tt.chunk_counts = c.t_chunk_counts;
  tt.supers = c.t_supers;
  tt.resolution[0] = c.cam.resolution[0];
  tt.resolution[1] = c.cam.resolution[1];
  tt.draw_cmd = tt.out = nullptr;
  tt.ticket = nullptr;
  tt.ticket_count = 0;
  TriEmitArgs& te = e.temit;
  te.tri_masks = c.tri_masks;
  te.visible = c.visible_out;
  te.vis = c.vis;
  te.tri_cmd = tri_cmd;
  te.chunk_counts = c.t_chunk_counts;
  te.supers = c.t_supers;
  te.draw_cmd = draw_cmd;
  te.out = c.reordered_out;
}

// ---- multi-view meshlet stage: oxc_cull_geometry_batch over several VIEWS of one scene (cull_meshes + cull_meshlets per view) ----
// One pass over the scene's meshlets instead of one per view: the views that kept a mesh instance at the same LOD form a group,
// a wave loads a 256-meshlet chunk of the group's bounds records once and tests it against every view of the group
// (cull_meshlets_hpb.slang:59-79 is the reference's own one-thread-many-views shape).  Per view the output is what the per-view
// kernels write: the view's own ascending list of indices into the view's own MeshletInstance list.
struct MvView {
  const InstCache* rows;         // this view's instance rows (prepare_body with this view's camera)
  const uint32_t* mesh_counts;   // [M] meshlets of the instance in this view's list (0: culled by this view's cull_meshes)
  const uint32_t* mesh_offsets;  // [M] first record of the instance in this view's list
  uint32_t* vchunks;             // [M] 256-meshlet chunks of the instance in this view ...
  uint32_t* vchunk0;             // [M] ... and their exclusive prefix: the view's own chunk numbering
  uint64_t* bits;                // [chunks][4] survivor ballots per view chunk
  uint32_t* counts;              // [chunks]
  uint32_t* idbase;              // [chunks] list index of the chunk's first meshlet
  uint32_t* supers;              // per 64 chunks (stride kSuperStride)
  uint32_t* scan_total;          // [2] number of chunks of this view
  uint32_t* out;                 // visible_meshlet_instances_indices of this view
  uint32_t* runs;                // [M][2] optional: {first, count} of the instance in this view's (possibly implicit) MeshletInstance list
  uint32_t* tri_cmd;
  uint32_t n_cap;                // the view's list capacity
  uint32_t n_supers;
  float cam_pos[3];
  uint32_t _pad;
};
struct MvGroup {  // the views of one mesh instance that read the same bounds array with the same transform and the same count
  uint64_t bounds;
  uint32_t count;
  uint32_t view_mask;
};
struct MvBlob {
  MvView v[kMaxBatch];
};
struct MvArgs {
  uint32_t views, M;
  uint32_t same_pos;  // every view has the same camera position: the normal cone is evaluated once per meshlet, not once per view
  MvView* dev;        // device copy of the table (written by k_mv_group)
  MvGroup* groups;    // [M][views]
  uint32_t* grp_chunks;  // [M] chunks of all groups of the instance
  uint32_t* inst_step0;  // [M] exclusive prefix
  uint32_t* step_total;  // [2]
  uint2* steps;          // [total] {mesh instance, group | chunk << 8}
  uint32_t* tickets;
};
void launch_mv_setup(const MvArgs& a, const MvBlob& blob, uint32_t grid, hipStream_t s);  // groups, chunk numbering, step list
void launch_mv_test(const MvArgs& a, uint32_t num_cus, hipStream_t s);
void launch_mv_emit(const MvArgs& a, uint32_t max_chunks_per_view, uint32_t max_grid, hipStream_t s);

struct HizArgs {
  const float* depth;
  float* hiz;
  uint32_t dw, dh;
  uint32_t w, h, levels;
  uint32_t level_off[13];  // floats
};

void launch_prepare(const PrepareArgs& a, uint32_t grid, uint32_t views, hipStream_t s);
void launch_hpb_test(const HpbTestArgs& a, uint32_t grid, hipStream_t s);
void launch_scan_mesh_counts(const uint32_t* counts, uint32_t* offsets, uint32_t n, uint32_t cap, uint32_t* vis, uint32_t* cmd, hipStream_t s);
void launch_expand(const uint32_t* counts, const uint32_t* offsets, uint32_t n, uint32_t cap, void* out, uint32_t grid, hipStream_t s);
// grid_limit != 0: at most that many blocks (async_triangles: the HiZ variants leave wave slots to the triangle stage running beside them)
// a.out != null: the unordered (appending) instantiation of the same kernel; no emit launch follows
void launch_meshlets_test(const MeshletTestArgs& a, bool hiz, bool occl, bool late, uint32_t grid, uint32_t num_cus, uint32_t grid_limit, hipStream_t s);
void launch_meshlets_emit(const MeshletEmitArgs& a, bool hiz, bool late, uint32_t grid, hipStream_t s);
void launch_tris_test(const TriTestArgs& a, bool late, bool wide, bool small_triangle_cull, bool cached_loads, uint32_t grid, hipStream_t s);  // cached_loads: shared (instanced) geometry -- plain loads instead of `nt`
void launch_tris_emit(const TriEmitArgs& a, bool late, uint32_t wide, uint32_t grid, hipStream_t s);  // wide: oxc_cull_geometry_context::wide_triangle_index (0, 1, 2 = pairs)
// unordered_output: test + expansion in one launch (a.draw_cmd / a.out set); grid in kFusedTriSpan-meshlet spans
void launch_tris_fused(const TriTestArgs& a, bool late, uint32_t wide, bool small_triangle_cull, bool cached_loads, uint32_t grid, uint32_t resident_cus, hipStream_t s);
void launch_hiz(const HizArgs& a, uint32_t num_cus, hipStream_t s);
// batched (grid.y = batch element); `dev` is the device copy written by launch_prepare_batch
void launch_prepare_batch(const BatchBlob& blob, BatchElem* dev, uint32_t grid, hipStream_t s);
void launch_scan_batch(const BatchElem* dev, uint32_t count, hipStream_t s);
void launch_expand_batch(const BatchElem* dev, uint32_t count, uint32_t grid, hipStream_t s);
void launch_meshlets_test_batch(const BatchElem* dev, uint32_t count, uint32_t grid, hipStream_t s);
void launch_meshlets_emit_batch(const BatchElem* dev, uint32_t count, uint32_t grid, hipStream_t s);
void launch_tris_test_batch(const BatchElem* dev, uint32_t count, uint32_t grid, hipStream_t s);
void launch_tris_emit_batch(const BatchElem* dev, uint32_t count, uint32_t grid, hipStream_t s);
void launch_seed_slot(uint32_t* slot, uint32_t total, hipStream_t s);
void launch_pack_counters(const uint32_t* vis, const uint32_t* tri_cmd, const uint32_t* draw_cmd, uint32_t* out4, hipStream_t s);
struct PackBlob {  // one element per context of a batched call (null pointers read as 0)
  const uint32_t* vis[kMaxBatch];
  const uint32_t* tri_cmd[kMaxBatch];
  const uint32_t* draw_cmd[kMaxBatch];
  uint32_t count;
};
void launch_pack_counters_batch(const PackBlob& blob, uint32_t* out4, hipStream_t s);
void launch_stream_read(const void* p, uint64_t bytes, uint32_t* sink, uint32_t grid, hipStream_t s);
void launch_debug_decode_bounds(const void* bounds, uint32_t n, float* out10, hipStream_t s);
struct DebugProjectArgs {
  float mvp[16];
  float near_clip;
  uint32_t n;
  const float* boxes6;
  float* out7;
};
void launch_debug_project_aabb(const DebugProjectArgs& a, hipStream_t s);
// oxcull_raster.hip: consumer of the indirect draw
struct TriSetup;
// What vs_main needs of a mesh instance, resolved once per draw by k_draw_rows (mesh_instance -> mesh -> lods[lod] is three dependent
// loads per vertex otherwise): the LOD's arrays and rows 0..2 of the world matrix.
struct DrawRow {
  uint64_t meshlets, micro, vidx, positions;
  float w[12];
};
struct DrawArgs {
  float pv[16];
  DrawRow* rows;
  uint32_t mesh_instance_count;
  const GpuMesh* meshes;
  const float* transforms;
  const GpuMeshInstance* mesh_instances;
  const GpuMeshletInstance* meshlet_instances;
  const uint32_t* indices;    // reordered_indices
  const uint32_t* draw_cmd;   // VkDrawIndexedIndirectCommand: [0] = indexCount
  unsigned long long* visdepth;
  uint32_t width, height;
  uint32_t wide;
  // Big triangles (pixel box beyond the small path) are queued in kBigSegs segments of the big list, each with its own counter
  // (stride kBigSegStride words): a single counter retires ~13 ns per wave-level atomic on this part, which serialised the setup kernel.
  uint32_t big_seg_capacity;  // triangles per segment
  TriSetup* big_list;
  uint32_t* big_seg_counts;
  uint32_t clip_capacity;  // triangles that cross a clip plane: ids queued for k_draw_clipped
  uint32_t* clip_list;
  uint32_t* clip_count;
  uint32_t tile_capacity;  // 64 x 64 pixel tiles of the big triangles' boxes: {big list index, tile} pairs, one wave each in k_draw_big
  uint2* tile_list;
  uint32_t* tile_count;
};
void launch_draw_visbuffer(const DrawArgs& a, bool clear, float* depth_out, uint32_t* vis_out, uint32_t max_grid, hipStream_t s);
constexpr uint32_t kTriSetupBytes = 40;
constexpr uint32_t kBigSegs = 256, kBigSegStride = 16;
constexpr uint32_t kRasterHeaderBytes = 256 + kBigSegs * kBigSegStride * 4;  // clip / tile counters, then the segment counters
// oxcull_terrain.hip: terrain patch cull
struct TerrainArgs {
  float pv[16];
  float near_clip;
  uint32_t cull_flags;
  float world_min[2], world_size[2];
  uint32_t pcx, pcy;
  float base_height, height_scale;
  const float2* patch_minmax;
  const float* hiz_data;
  uint32_t hiz_level_off[13];
  uint32_t hiz_w, hiz_h, hiz_levels;
  uint32_t* mask;
  uint32_t* visible;
  uint32_t* draw_cmd;
  // scratch of the two-kernel form (more than 1024 patches): one emit ballot per wave, one count per 1024-patch block
  uint64_t* emit_bits;
  uint32_t* block_counts;
};
void launch_cull_terrain(const TerrainArgs& a, hipStream_t s);
// oxcull_hpb.hip: hierarchical page buffer producer
void launch_generate_hpb(const uint32_t* page_table, uint8_t* data, uint32_t w, uint32_t h, uint32_t layers, uint32_t levels, const uint64_t* level_offset,
                         hipStream_t s);
// oxcull_bounds.hip: meshlet bounds producer
void launch_build_meshlet_bounds(const float* pos, uint32_t vertex_count, const void* meshlets, uint32_t meshlet_count, const uint32_t* vidx,
                                 const uint8_t* micro, void* out_bounds, float* out_mesh6, void* out_qpos, float* meshlet_minmax, float* normals,
                                 uint32_t* normal_counts, float* fold_scratch, uint32_t chunk, uint32_t max_grid, hipStream_t s);

void launch_quantize_vertex_streams(const float* pos, const float* nrm, const float* uv, uint32_t vertex_count, void* out_qpos, void* out_qnrm,
                                    void* out_quv, uint32_t max_grid, hipStream_t s);
}  // namespace oxc
