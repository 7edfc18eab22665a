/*
 * oxcull_oracle.h -- CPU oracle for the Oxylus meshlet visibility pipeline.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (oxylus_amd/, include/) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker.
 *
 * PARITY UNPINNED: the reference (oxylusengine/Oxylus) ships no golden vectors, KATs or
 * tests for this path and its Slang->SPIR-V->Vulkan path cannot
 * be built or run in this image (no slangc, no Vulkan ICD, no xmake, un-vendored vuk/glm).
 * This file is therefore a restatement of the reference shaders' arithmetic with a fixed
 * canonical IEEE-754 binary32 evaluation order: round-to-nearest,
 * no FMA contraction (compile with -ffp-contract=off), left-to-right dot / mat*vec,
 * correctly rounded sqrt and divide.  Output lists are emitted in ascending order (the
 * reference's order is atomic-race dependent; compare as sorted sets against it).
 *
 * All struct layouts are the reference's GPU layouts (scalar layout, little endian):
 * Oxylus/include/Scene/SceneGPU.hpp:84-152,222-229.
 */
#ifndef OXCULL_ORACLE_H
#define OXCULL_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Oxylus/include/Scene/SceneGPU.hpp:84-90 */
typedef struct {
  uint16_t aabb_center[3];
  int8_t cone_axis_xy[2];
  uint16_t aabb_extent[3];
  int8_t cone_axis_z;
  int8_t cone_cutoff;
} orc_meshlet_bounds; /* 16 B */

/* SceneGPU.hpp:105-108 */
typedef struct {
  uint32_t mesh_instance_index;
  uint32_t meshlet_index;
} orc_meshlet_instance; /* 8 B */

/* SceneGPU.hpp:110-116 */
typedef struct {
  uint32_t mesh_index;
  uint32_t lod_index;
  uint32_t material_index;
  uint32_t transform_index;
  uint32_t meshlet_instance_visibility_offset;
} orc_mesh_instance; /* 20 B */

/* SceneGPU.hpp:118-123 */
typedef struct {
  uint32_t indirect_vertex_index_offset;
  uint32_t local_triangle_index_offset; /* BYTE offset into the u8 stream */
  uint32_t vertex_count;
  uint32_t triangle_count;
} orc_meshlet; /* 16 B */

/* SceneGPU.hpp:125-139 (pointers are addresses valid in the caller's address space) */
typedef struct {
  uint64_t indices;
  uint64_t meshlets;
  uint64_t meshlet_bounds;
  uint64_t local_triangle_indices;
  uint64_t indirect_vertex_indices;
  uint32_t indices_count;
  uint32_t meshlet_count;
  uint32_t meshlet_bounds_count;
  uint32_t local_triangle_indices_count;
  uint32_t indirect_vertex_indices_count;
  float error;
} orc_mesh_lod; /* 64 B */

/* SceneGPU.hpp:141-152 */
typedef struct {
  uint64_t vertex_positions; /* u16x4, stride 8 */
  uint64_t vertex_normals;
  uint64_t texture_coords;
  uint32_t vertex_count;
  uint32_t lod_count;
  uint64_t lods;
  float aabb_center[3];
  float aabb_extent[3];
} orc_mesh; /* 64 B */

/* SceneGPU.hpp:222-229 -- 96 B push constant */
typedef struct {
  float projection_view[16]; /* column-major: element (r,c) at [c*4+r] */
  float position[3];
  float acceptable_lod_error;
  float resolution[2];
  float near_clip;
  uint32_t mesh_instance_count;
} orc_cull_camera;

/* SceneGPU.hpp:97-104 */
typedef struct {
  uint32_t total_visible_meshlet_instances;
  uint32_t early_visible_meshlet_instances;
  uint32_t late_visible_meshlet_instances;
} orc_visibility;

/* Linear R32F mip chain standing in for the vuk ImageAttachment. */
typedef struct {
  const float* data;
  uint32_t width, height, levels;
  uint64_t level_offset[13]; /* in floats */
} orc_hiz;

enum {
  ORC_TEST_FRUSTUM = 1u, /* SceneGPU.hpp:345-353 */
  ORC_SELECT_LOD = 2u,
  ORC_TEST_OCCLUSION = 4u,
  ORC_LATE_PASS = 8u,
  ORC_TEST_ALL = 7u
};

/* Decision-margin statistics ("boundary set"): elements with at least one
 * comparison whose two sides are within 4 ulp of each other. */
typedef struct {
  uint64_t meshlets_near_threshold;
  uint64_t triangles_near_threshold;
} orc_margin_stats;

/* ---- scalar functions (cull.slang, common/math.slang, scene.slang) ---- */
float orc_dequantize_half(uint16_t h);
void orc_mul_mat4(const float* a, const float* b, float* out);
int orc_test_frustum(const float* mvp, const float* center, const float* extent);
int orc_test_cone(const float* center, float radius, const float* axis, float cutoff, const float* cam);
int orc_project_aabb(const float* mvp, float near_clip, const float* center, const float* extent, float* out6);
int orc_test_occlusion(const float* screen_aabb6, const orc_hiz* hiz);
uint32_t orc_occlusion_mip(const float* screen_aabb6, const orc_hiz* hiz);
float orc_sample_level_min_reduction_2x2(const orc_hiz* hiz, float u, float v, uint32_t mip);
int orc_test_triangle_backface(const float* clip3x4);
void orc_normal_matrix(const float* world, float* out9);
float orc_to_world_radius(const float* world, float radius);
void orc_decode_bounds(const orc_meshlet_bounds* b, float* center, float* extent, float* axis, float* cutoff);

/* ---- kernels ---- */
/* passes/hiz.slang + Passes/CullGeometry.cpp:10-59.  hiz->data is written. */
void orc_generate_hiz(const float* depth, uint32_t depth_w, uint32_t depth_h, orc_hiz* hiz);

/* passes/cull_meshes.slang:17-85.  Deterministic: instances expanded in ascending
 * mesh-instance order.  Returns total meshlet instances; writes lod_index. */
uint32_t orc_cull_meshes(const orc_mesh* meshes, const float* transforms, orc_mesh_instance* mesh_instances,
                         const orc_cull_camera* cam, uint32_t cull_flags, orc_meshlet_instance* meshlet_instances_out,
                         uint32_t* cull_meshlets_cmd3);

/* passes/cull_meshlets.slang:23-73 on [begin,end).  Appends to out (ascending). Returns count. */
uint32_t orc_cull_meshlets(const orc_mesh* meshes, const float* transforms, const orc_mesh_instance* mesh_instances,
                           const orc_meshlet_instance* meshlet_instances, uint32_t begin, uint32_t end,
                           const orc_cull_camera* cam, uint32_t* visible_out, orc_margin_stats* stats);

/* Same, split over nthreads contiguous ranges (pthread); output concatenated in order. */
uint32_t orc_cull_meshlets_mt(const orc_mesh* meshes, const float* transforms, const orc_mesh_instance* mesh_instances,
                              const orc_meshlet_instance* meshlet_instances, uint32_t total,
                              const orc_cull_camera* cam, uint32_t* visible_out, uint32_t nthreads);

/* Same result; every thread repeats its range `passes` times (CPU-baseline timing without
 * paying thread start-up per pass). */
uint32_t orc_cull_meshlets_mt_passes(const orc_mesh* meshes, const float* transforms, const orc_mesh_instance* mesh_instances,
                                     const orc_meshlet_instance* meshlet_instances, uint32_t total,
                                     const orc_cull_camera* cam, uint32_t* visible_out, uint32_t nthreads, uint32_t passes);

/* passes/cull_meshlets_hiz.slang:19-88.  vis->early/late updated, mask updated in place,
 * visible_out written at [0,early) (early pass) or [early, early+late) (late pass).
 * Returns number emitted by this pass (= cull_triangles_cmd.x). */
uint32_t orc_cull_meshlets_hiz(const orc_mesh* meshes, const float* transforms, const orc_mesh_instance* mesh_instances,
                               const orc_meshlet_instance* meshlet_instances, const orc_cull_camera* cam,
                               uint32_t cull_flags, const orc_hiz* hiz, orc_visibility* vis, uint32_t* mask,
                               uint32_t* visible_out, orc_margin_stats* stats);

/* GPU::VirtualClipmap (SceneGPU.hpp:335-339), 76 B */
typedef struct {
  float projection_view_mat[16];
  int32_t page_offset[2];
  float z_near;
} orc_virtual_clipmap;

/* R8UI Texture2DArray with mips (the hierarchical page buffer), linear: level k holds
 * `layers` planes of max(1,w>>k) x max(1,h>>k) bytes at data + level_offset[k]. */
typedef struct {
  const uint8_t* data;
  uint32_t width, height, layers, levels;
  uint64_t level_offset[13]; /* bytes */
} orc_hpb;

int orc_test_vsm_page(const float* screen_aabb6, const orc_hpb* hpb, uint32_t layer, const int32_t* page_offset2);

/* passes/cull_meshlets_hpb.slang:25-99 (VSM multi-view: visible if ANY dirty clipmap view sees it).
 * cam->position carries -light_dir (Shadowmaps.cpp:433-455).  Returns the count; ascending. */
uint32_t orc_cull_meshlets_hpb(const orc_mesh* meshes, const float* transforms, const orc_mesh_instance* mesh_instances,
                               const orc_meshlet_instance* meshlet_instances, uint32_t total, const orc_cull_camera* cam,
                               const orc_virtual_clipmap* clipmaps, const uint32_t* clipmap_dirty_flags, uint32_t clipmap_count,
                               const orc_hpb* hpb, uint32_t* visible_out);

/* passes/cull_triangles.slang:27-90 over `count` visible slots starting at slot `first`.
 * Returns index_count (3 * passing triangles); reordered_out gets packed indices,
 * meshlets in slot order, triangles ascending within a meshlet. */
uint32_t orc_cull_triangles(const orc_mesh* meshes, const float* transforms, const orc_mesh_instance* mesh_instances,
                            const orc_meshlet_instance* meshlet_instances, const uint32_t* visible, uint32_t first,
                            uint32_t count, const orc_cull_camera* cam, uint32_t* reordered_out,
                            orc_margin_stats* stats);

/* Extension (no reference behaviour): meshlets with up to 128 triangles and the wide
 * packed index (meshlet_instance_index << 9) | (t*3+k).  Same per-triangle test. */
uint32_t orc_cull_triangles_wide(const orc_mesh* meshes, const float* transforms, const orc_mesh_instance* mesh_instances,
                                 const orc_meshlet_instance* meshlet_instances, const uint32_t* visible, uint32_t first,
                                 uint32_t count, const orc_cull_camera* cam, uint32_t* reordered_out);

/* The opt-in small-triangle cull (include/oxcull.h: oxc_cull_geometry_context::small_triangle_cull; no reference
 * behaviour, the north star names it).  orc_test_triangle_small: 1 = dropped. */
int orc_test_triangle_small(const float* clip3x4, const float* resolution2);
/* wide: include/oxcull.h wide_triangle_index -- 0 = 24 + 8 bit packed index, 1 = 23 + 9 bits, 2 = SURVEY A.7's pairs
 * {u32 meshlet_instance_index, u32 t*3+k}: reordered_out then holds two words per index, the return value stays the index count. */
uint32_t orc_cull_triangles_flags(const orc_mesh* meshes, const float* transforms, const orc_mesh_instance* mesh_instances,
                                  const orc_meshlet_instance* meshlet_instances, const uint32_t* visible, uint32_t first,
                                  uint32_t count, const orc_cull_camera* cam, uint32_t* reordered_out, int wide,
                                  int small_triangle_cull);
/* Conditioning-aware boundary set of cull_triangles (see the .c): flags64[s * 64 + t] = 1 when the decision of triangle t of
 * visible slot first + s can be flipped by a different legal evaluation order of the same formula. */
void orc_triangle_boundary_flags(const orc_mesh* meshes, const float* transforms, const orc_mesh_instance* mesh_instances,
                                 const orc_meshlet_instance* meshlet_instances, const uint32_t* visible, uint32_t first,
                                 uint32_t count, const orc_cull_camera* cam, uint8_t* flags64);
/* 1 when this library is the -DORC_FAST_ENVELOPE build (liboxcull_oracle_fast.so), see oxcull_oracle.c */
int orc_is_fast_envelope(void);

uint32_t orc_cull_triangles_mt(const orc_mesh* meshes, const float* transforms, const orc_mesh_instance* mesh_instances,
                               const orc_meshlet_instance* meshlet_instances, const uint32_t* visible, uint32_t first,
                               uint32_t count, const orc_cull_camera* cam, uint32_t* reordered_out, uint32_t nthreads);

/* Config 1 harness: Scene.cpp:1690-1740 world-matrix chain + BoundingVolume.cpp:32-88. */
uint32_t orc_entities_update_and_cull(uint32_t n, const float* trs10 /* t3 q4 s3 */, const int32_t* parent,
                                      const float* aabb_min_max6, const float* frustum_planes24, float* world_out16,
                                      uint8_t* visible_out);
uint32_t orc_entities_update_and_cull_passes(uint32_t n, const float* trs10, const int32_t* parent, const float* aabb_min_max6,
                                             const float* frustum_planes24, float* world_out16, uint8_t* visible_out, uint32_t passes);

/* ---- meshlet bounds producer (asset side) --------------------------------------
 * Restates the loop body of Oxylus/src/Asset/AssetManager_GLTF.cpp:573-578 (position quantisation) and
 * :683-744 (per-meshlet AABB -> half, normal cone -> s8, mesh AABB).  The cone comes from a third-party
 * dependency that is NOT under /root/reference: meshoptimizer v1.2 (xmake/packages.lua:9), functions
 * meshopt_computeMeshletBounds -> meshopt_computeClusterBounds -> computeBoundingSphere (src/clusterizer.cpp)
 * and meshopt_quantizeHalf / meshopt_quantizeSnorm (src/meshoptimizer.h).  Its published algorithm is
 * restated here from the public sources (the bounding-sphere routine with per-axis extremum seeding and one
 * refinement sweep, as published since v0.22); v1.2 itself is not available in this image to diff against,
 * so this row is PARITY UNPINNED twice over.  Only the quantities the engine keeps (cone_axis_s8,
 * cone_cutoff_s8) are produced; the cluster sphere / apex, which the engine discards, are not. */
uint16_t orc_quantize_half(float v);
int orc_quantize_snorm(float v, int bits);
/* positions: float3 (stride 12).  out_bounds[meshlet_count]; out_mesh6 = {center xyz, extent xyz};
 * out_qpos (may be NULL) = u16x4 per vertex. */
void orc_build_meshlet_bounds(const float* positions, uint32_t vertex_count, const orc_meshlet* meshlets, uint32_t meshlet_count,
                              const uint32_t* indirect_vertex_indices, const uint8_t* local_triangle_indices,
                              orc_meshlet_bounds* out_bounds, float* out_mesh6, uint16_t* out_qpos);

/* The three vertex streams of AssetManager_GLTF.cpp:570-588: positions -> u16x4 halfs (w = 0), normals ->
 * ((snorm10(x)+511) << 20) | ((snorm10(y)+511) << 10) | (snorm10(z)+511), texcoords -> u16x2 halfs.
 * Any input may be NULL (that stream is skipped). */
void orc_quantize_vertex_streams(const float* positions, const float* normals, const float* texcoords, uint32_t vertex_count,
                                 uint16_t* out_qpos, uint32_t* out_qnrm, uint16_t* out_quv);

/* ---- HPB producer.  passes/rmvsm_downsample_hpb.slang:10-33 driven by
 * Passes/Shadowmaps.cpp:331-366; page flags rmvsm.slang:16-28 ([Flags]: Visible 1, Dirty 2, Backed 4). */
void orc_generate_hpb(const uint32_t* page_table, orc_hpb* hpb);

/* ---- terrain patch cull (passes/terrain_cull.slang:17-83, Passes/Terrain.cpp:159-216).
 * terrain8 = {world_min.xy, world_size.xy, base_height, height_scale, (float)patch_count.x, (float)patch_count.y}
 * (patch counts passed separately as integers too).  Returns the number of emitted patches (ascending). */
uint32_t orc_cull_terrain(const float* world_min2, const float* world_size2, uint32_t patch_count_x, uint32_t patch_count_y, float base_height,
                          float height_scale, const float* patch_minmax /* 2 floats per patch */, const orc_cull_camera* cam, uint32_t cull_flags,
                          const orc_hiz* hiz, uint32_t* mask, uint32_t* out_visible);

/* ---- consumer of the indirect draw ------------------------------------------------
 * What draw_for_visbuffer does with cull_geometry's outputs (Passes/DrawGeometry.cpp:104-190, pipeline
 * visbuffer_encode: vs_main passes/visbuffer_encode.slang:24-49, cullMode eBack, depth GreaterOrEqual,
 * reversed Z), as a software rasteriser with stated rules, because the fixed-function rasteriser's exact
 * sample rules are not available to match:
 *   vertex: VisBufferData(index) -> (meshlet instance, corner); Meshlet::index / Mesh::decode_position;
 *           world = mul(world, (p,1)).xyz; clip = mul(projection_view, (world,1))   (two steps, as vs_main);
 *   setup:  triangles with any clip.w <= 0 or a screen coordinate beyond +-2^20 pixels are dropped (no clipper);
 *           screen = (clip.xy / clip.w * 0.5 + 0.5) * extent, snapped to 1/256 pixel (nearest);
 *           back faces (signed fixed-point area >= 0, the sign cull_triangles' determinant test uses) dropped;
 *   cover:  pixel centres, integer edge functions, top-left rule;
 *   depth:  z/w interpolated in binary64: (e0 z0 + e1 z1 + e2 z2) * (1 / area) with the exact integer edge values, rounded to binary32; fragments outside
 *           (0, 1] dropped; per pixel the maximum of (depth bits << 32) | vis wins, vis = (instance << 8) |
 *           (corner / 3) (VisBufferData::encode) -- order-independent, i.e. one of the results the
 *           reference's race between equal-depth fragments can produce.
 * visdepth: u64[h * w], cleared to 0 by the caller before the first draw of a frame. */
void orc_draw_visbuffer(const orc_mesh* meshes, const float* transforms, const orc_mesh_instance* mesh_instances,
                        const orc_meshlet_instance* meshlet_instances, const uint32_t* indices, uint32_t index_count, const float* projection_view,
                        uint32_t width, uint32_t height, uint32_t corner_bits, uint64_t* visdepth);
void orc_resolve_visbuffer(const uint64_t* visdepth, uint32_t width, uint32_t height, float* depth, uint32_t* vis);

#ifdef __cplusplus
}
#endif
#endif
