/*
 * oxcull.h -- C ABI of the MI355X-native meshlet visibility pipeline (liboxcull.so).
 *
 * Drop-in boundary for the Oxylus engine's compute-path cull.  The engine has no plugin/FFI
 * table for this path; the boundary is the two RendererInstance members
 *
 *   auto generate_hiz(this RendererInstance&, MainGeometryContext&) -> void;
 *   auto cull_geometry(this RendererInstance&, CullGeometryContext&) -> void;
 *       (Oxylus/include/Render/RendererInstance.hpp:397-398, bodies in
 *        Oxylus/src/Render/Passes/CullGeometry.cpp:10-59 and :61-404)
 *
 * plus their context structs (RendererInstance.hpp:143-216).  This header is the plain-C
 * surface under a C++ shim that keeps those names (oxylus_amd/host/RendererInstance.hpp):
 * every vuk::Value<vuk::Buffer> becomes {device pointer, bytes}, every
 * vuk::Value<vuk::ImageAttachment> becomes a linear mip chain in device memory.  All buffers
 * use the reference's GPU byte layouts (Oxylus/include/Scene/SceneGPU.hpp:84-152,222-229), so a
 * Vulkan consumer could bind the outputs unchanged.
 *
 * Conventions: POD structs only, no exceptions cross the ABI, every entry point returns an
 * oxc_status.  A context is externally synchronised (the reference calls these from the main thread
 * only, RenderContext.cpp:585-586) and owns ONE set of scratch buffers (instance cache, survivor
 * bitmaps, chunk counts), so its calls are ordered: a call on a different hipStream_t than the
 * context's previous call first waits (hipStreamWaitEvent, on the device) for that previous call.
 * (Not across a HIP-graph capture: while hip_stream is being captured that wait is skipped -- work the context still has in flight
 * on another stream must be complete, or ordered by the caller's own events inside the capture, before the capture begins; and a
 * stream the context was last used on must outlive the next call on a different stream, or be synchronised before it is destroyed.)
 * Independent work that should overlap -- a main view and a shadow view, several frames in flight
 * -- uses one context per stream.  All work is enqueued asynchronously on the caller's hipStream_t
 * (passed as void*).  Entry points that synchronise the host: oxc_read_counters (and the test / measurement
 * hooks of oxcull_debug.h: oxc_debug_read_u32, oxc_profile_end), and any call that has to GROW scratch memory (oxc_reserve up front avoids that;
 * while the stream is being captured into a HIP graph a call that would have to grow returns
 * OXC_INVALID_ARG instead).
 */
#ifndef OXCULL_H
#define OXCULL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OXC_ABI_VERSION 5u

typedef struct oxc_ctx oxc_ctx;

typedef enum oxc_status {
  OXC_OK = 0,
  OXC_INVALID_ARG = 1,
  OXC_HIP_ERROR = 2,
  OXC_RCCL_ERROR = 3,
  OXC_OUT_OF_MEMORY = 4
} oxc_status;

/* GPU::CullFlag, SceneGPU.hpp:345-353 (spec constant 0 of every cull pipeline,
 * CullGeometry.cpp:89,156,298,363).  HAS_FLAG(mask, a|b) in the shaders means "any of". */
enum {
  OXC_CULL_NONE = 0u,
  OXC_CULL_TEST_FRUSTUM = 1u << 0,
  OXC_CULL_SELECT_LOD = 1u << 1,
  OXC_CULL_TEST_OCCLUSION = 1u << 2,
  OXC_CULL_LATE_PASS = 1u << 3,
  OXC_CULL_TEST_ALL = 7u
};

/* Which stages of cull_geometry to run (extension; 0 = all, as the reference always does).
 * Used by the benchmark for the "frustum+cone cull only" configuration. */
enum {
  OXC_STAGE_MESHES = 1u << 0,    /* cull_meshes   (only honoured when init_cull_meshes) */
  OXC_STAGE_MESHLETS = 1u << 1,  /* cull_meshlets / cull_meshlets_hiz */
  OXC_STAGE_TRIANGLES = 1u << 2, /* cull_triangles */
  OXC_STAGE_ALL = 7u
};

/* vuk::Value<vuk::Buffer> stand-in: device pointer + size. */
typedef struct oxc_buffer {
  void* dptr;
  uint64_t bytes;
} oxc_buffer;

/* vuk::Value<vuk::ImageAttachment> stand-in for R32F / D32F images: a linear, row-major mip
 * chain in one device allocation.  Level k is max(1,width>>k) x max(1,height>>k) floats at
 * byte offset level_offset[k]. */
typedef struct oxc_image {
  void* dptr;
  uint32_t width, height, levels, _pad;
  uint64_t level_offset[13]; /* bytes; hiz.slang binds at most 13 mips (CullGeometry.cpp:24,36-38) */
} oxc_image;

/* R8UI Texture2DArray with mips (the VSM hierarchical page buffer, `hpb_attachment`): level k
 * holds `layers` planes of max(1,width>>k) x max(1,height>>k) bytes at byte offset level_offset[k]. */
typedef struct oxc_image_array_u8 {
  void* dptr;
  uint32_t width, height, layers, levels;
  uint64_t level_offset[13];
} oxc_image_array_u8;

/* GPU::VirtualClipmap (SceneGPU.hpp:335-339), 76 bytes: the element type of vsm_clipmaps_buffer. */
typedef struct oxc_virtual_clipmap {
  float projection_view_mat[16];
  int32_t page_offset[2];
  float z_near;
} oxc_virtual_clipmap;

/* GPU::CullCamera -- 96 B push constant (SceneGPU.hpp:222-229, scene.slang:196-203). */
typedef struct oxc_cull_camera {
  float projection_view[16]; /* glm::mat4, column-major */
  float position[3];
  float acceptable_lod_error;
  float resolution[2];
  float near_clip;
  uint32_t mesh_instance_count;
} oxc_cull_camera;

/* The PreparedFrame buffers the cull path touches (RendererInstance.hpp:143-169; sizes from
 * RendererInstance.cpp:1640-1732).  Caller-owned device memory. */
typedef struct oxc_prepared_frame {
  uint32_t mesh_instance_count;
  uint32_t max_meshlet_instance_count;
  oxc_buffer meshes_buffer;                           /* GPU::Mesh[]            read  */
  oxc_buffer transforms_world_buffer;                 /* GPU::TransformWorld[]  read  */
  oxc_buffer mesh_instances_buffer;                   /* GPU::MeshInstance[]    read; lod_index written by cull_meshes */
  oxc_buffer meshlet_instances_buffer;                /* GPU::MeshletInstance[] written by cull_meshes, read after */
  oxc_buffer visible_meshlet_instances_indices_buffer; /* u32[max_meshlet_instance_count] written */
  oxc_buffer meshlet_instance_visibility_mask_buffer; /* u32[ceil(N/32)] persistent, read+written when use_hiz */
  oxc_buffer reordered_indices_buffer;                /* u32[N*64*3] written by cull_triangles */
} oxc_prepared_frame;

/* CullGeometryContext (RendererInstance.hpp:171-197).  Field names are the reference's. */
typedef struct oxc_cull_geometry_context {
  uint32_t struct_size; /* sizeof(oxc_cull_geometry_context), for ABI evolution */
  uint32_t use_hiz;     /* cull_meshlets_hiz path (two-pass occlusion) */
  uint32_t use_hpb;     /* cull_meshlets_hpb path (VSM multi-view page cull) */
  uint32_t init_cull_meshes;
  uint32_t cull_flags;  /* OXC_CULL_* */
  uint32_t stages;      /* OXC_STAGE_*; 0 = all */
  oxc_cull_camera cull_camera;
  oxc_image hiz_attachment; /* read when use_hiz */
  /* read when use_hpb (RendererInstance.hpp:183-192, Shadowmaps.cpp:433-463) */
  oxc_image_array_u8 hpb_attachment;
  oxc_buffer vsm_clipmaps_buffer;            /* oxc_virtual_clipmap[vsm_clipmap_count] */
  oxc_buffer vsm_clipmap_dirty_flags_buffer; /* u32[vsm_clipmap_count] */
  uint32_t vsm_clipmap_count;                /* <= 16 */
  /* Extension (no reference behaviour): 0 = the reference's packed index (id << 8) | (3t+k),
   * 64 triangles per meshlet; 1 = wide index (id << 9) | (3t+k) for meshlets of up to 128 triangles
   * (at most 2^23 meshlet instances per call, reordered_indices_buffer >= N*128*3*4 bytes);
   * 2 = SURVEY A.7's form for larger shards: every index is the PAIR {u32 meshlet_instance_index, u32 3t+k}
   * (8 bytes, little endian, id first), meshlets of up to 128 triangles, no id limit below 2^32,
   * reordered_indices_buffer >= N*128*3*8 bytes, same order of entries as the packed forms.  The decode of
   * visbuffer.slang:9-14 becomes instance = pair.x, corner = pair.y (oxc_draw_visbuffer does that).
   * DrawIndexedIndirect.index_count stays the number of INDICES (3 per triangle) and stays a u32: a call that
   * emits more than 2^32 - 1 of them (possible only when N*384 >= 2^32, i.e. N > 11 184 810, and then only if
   * nearly every triangle passes) sets instanceCount = 0 -- the command draws nothing -- and oxc_read_counters
   * of that call returns OXC_INVALID_ARG; every write stays inside the buffer. */
  uint32_t wide_triangle_index;
  /* Extension named by the north star ("per-triangle backface + small-triangle cull"); the reference has only the
   * clip-z and backface tests (cull_triangles.slang:68-69, cull.slang:169-171).  0 (default) = reference behaviour,
   * output byte-identical to a build without the flag.  1 = a triangle that passed both reference tests is ALSO
   * dropped when its screen-space bounding box covers no pixel centre of a cull_camera.resolution target:
   *   per corner (only when all three clip.w > 0, otherwise the triangle is kept):
   *     s.x = ((clip.x / clip.w) * 0.5 + 0.5) * resolution.x,  s.y likewise     (IEEE binary32, no contraction)
   *   lo = min over corners, hi = max over corners (per axis)
   *   dropped iff floor(lo.x + 0.5) == floor(hi.x + 0.5) || floor(lo.y + 0.5) == floor(hi.y + 0.5)
   * (pixel centres sit at k + 0.5: an interval [lo, hi] holds one iff the two roundings differ).  Not supported by
   * the fused path of oxc_cull_geometry_batch (such elements are processed one after the other). */
  uint32_t small_triangle_cull;
  /* EXPERIMENTAL extension (scheduling only, no effect on any output byte; on MI355X it has not been faster than the in-order call in any
   * measured configuration -- both stages keep the vector ALUs and the memory system busy, DESIGN.md 4c -- and exists for engines whose
   * draw sits between the two calls): 1 = the triangle stage of this call (cull_triangles test + ordered
   * emit) is enqueued on a second stream the context owns and runs BESIDE whatever is enqueued on hip_stream next -- typically the
   * meshlet stage of the following oxc_cull_geometry call (ALU-bound, while the triangle stage is HBM-bound) and oxc_generate_hiz.
   * reordered_indices_buffer and draw_geometry_cmd_buffer of this call are complete on a stream only after
   * oxc_join_triangles(ctx, that stream); every oxc_* entry point that reads them (oxc_read_counters, oxc_pack_counters,
   * oxc_draw_visbuffer) joins by itself.  The context orders everything it owns or writes: a later call waits where it would
   * overwrite what a pending triangle stage still reads (the visible list, MeshletInstance records rewritten by cull_meshes, its own
   * scratch).  0 (default) = the whole call is in order on hip_stream, as the reference records it. */
  uint32_t async_triangles;
  /* Extension (caching only, no effect on any output byte): the two HiZ calls of a frame -- early, then OXC_CULL_LATE_PASS, as
   * RendererInstance.cpp:842-884 records them -- run the same frustum and normal-cone tests on the same operands (the camera, the
   * transforms and the MeshletInstance list do not change between them; only the pyramid and the mask do).  With 1 on BOTH calls the
   * early call also evaluates the cone for the meshlets that were not visible last frame and leaves one "passed frustum and cone" bit
   * per meshlet in the context's scratch, and the late call reads those bits instead of testing again (its meshlets outside the
   * frustum are not even fetched).  By setting it on the late call the caller states that nothing those tests read has been written
   * since the early call: meshes / transforms / mesh instances / MeshletInstance list / meshlet bounds.  What the context can
   * check it checks -- the late call reuses the bits only if the previous flagged early call on this context had the same cull_camera
   * (all 96 bytes), the same buffers, counts and flags, and no call in between rebuilt the list (init_cull_meshes) or the scratch;
   * otherwise it silently tests again.  (The match is made when the call is enqueued: a late call captured into a HIP graph on its own
   * reuses, at every replay, the bits of whatever flagged early call ran last -- capture the pair, or keep the scene unchanged
   * between the replays.)  An early call that is in order on one stream (no async_triangles) also does the prepare work of the late call
   * that follows it directly (same capture, if any), which then launches no prepare kernel.  Only with use_hiz
   * and OXC_CULL_TEST_OCCLUSION; ignored elsewhere and by oxc_cull_geometry_batch.  0 (default) = every call tests on its own. */
  uint32_t share_pass_tests;
  /* Extension (output ORDER only; counts and the SET of emitted ids / packed triangles are those of the ordered form, and the mask bytes
   * are identical): how the compacted lists are laid out.  The reference allocates output slots with atomics -- one atomic_add per
   * 64-thread workgroup in cull_meshlets.slang:55-70 and cull_triangles.slang:71-88, two per visible thread in
   * cull_meshlets_hiz.slang:67-78 -- so its order is whatever the race gives.
   *   0 (default) = ascending lists, deterministic: test -> ballots -> ordered emit, two launches per stage.
   *   1 = unordered where that is the faster form on this part: the triangle stage is ONE launch (a block tests a span of visible
   *       meshlets -- implementation-defined, currently 128 -- and appends its packed indices behind one atomic_add on index_count);
   *       the plain meshlet stage (no use_hiz / use_hpb) is one launch (one atomic_add on cull_triangles_cmd.x per 1024 meshlets).
   *       The HiZ / HPB meshlet stages keep the ordered two-launch form and their ascending visible list.
   *   Any other value returns OXC_INVALID_ARG.  (Rounds 4 built "2": the HiZ meshlet tests appending with one atomic_add pair per
   *   256-meshlet wave step, the reference's literal scheme aggregated through the ballot -- 150 / 154 us per launch against 79 + 11 /
   *   66 + 11 for test + ordered emit, every step queueing on two addresses that retire ~88 atomics per microsecond; removed in round 5,
   *   the measurement is in DESIGN.md.)
   * Inside a block's run the ids ascend; the runs land in arrival order.  A triangle's three packed indices stay adjacent.  Sorting a
   * list gives the bytes of the ordered form (tests/test_gpu_unordered.py).  Ignored by oxc_cull_geometry_batch's fused path. */
  uint32_t unordered_output;
  /* Extension (configs[4]: many views of one scene): 1 = cull_meshes leaves this view's MeshletInstance list IMPLICIT instead of writing
   * one 8-byte record per meshlet and view (466 MB per 16 cascade views of a 10 M-meshlet scene, a third of the call): record i of the
   * view is {mesh instance m, meshlet i - first[m]} for first[m] <= i < first[m] + count[m], with {first[m], count[m]} written to
   * meshlet_instance_runs_buffer (u32[2 * mesh_instance_count]; count 0 = the view's cull_meshes dropped the instance) -- "emit runs,
   * expand in the consumer".  visible_meshlet_instances_indices, the counters and lod_index are exactly those of the explicit form;
   * expanding the runs gives the explicit list byte for byte (tests/test_gpu_round2.py).  Only oxc_cull_geometry_batch's multi-view path
   * has no reader of the records (its meshlet test walks the instances' bounds directly), so only there is the flag accepted: every
   * element must set it, stages must not include OXC_STAGE_TRIANGLES; any other call with the flag returns OXC_INVALID_ARG.  The
   * runs buffer alone (flag 0) is also filled by that path when given. */
  uint32_t implicit_meshlet_instances;
  uint32_t _reserved1; /* must be 0 */
  /* in/out: produced when init_cull_meshes, consumed (and updated) by later calls of the
   * sequence, exactly like the reference's hoisted context (RendererInstance.cpp:793-800). */
  oxc_buffer visibility_buffer;        /* GPU::MeshletInstanceVisibility {total, early, late} */
  oxc_buffer cull_meshlets_cmd_buffer; /* VkDispatchIndirectCommand {x, 1, 1} */
  /* out: fresh per call (CullGeometry.cpp:125-127, 380-382) */
  oxc_buffer cull_triangles_cmd_buffer; /* VkDispatchIndirectCommand {#visible meshlets, 1, 1} */
  oxc_buffer draw_geometry_cmd_buffer;  /* VkDrawIndexedIndirectCommand {indexCount, 1, 0, 0, 0} */
  /* out, optional, caller-owned: {first, count} of every mesh instance in this view's MeshletInstance list (see implicit_meshlet_instances) */
  oxc_buffer meshlet_instance_runs_buffer;
} oxc_cull_geometry_context;

/* MainGeometryContext fields used by generate_hiz (RendererInstance.hpp:199-216). */
typedef struct oxc_main_geometry_context {
  uint32_t struct_size;
  uint32_t _pad;
  oxc_image depth_attachment; /* levels = 1; read */
  oxc_image hiz_attachment;   /* written: all `levels` mips */
} oxc_main_geometry_context;

typedef struct oxc_counters {
  uint32_t total_visible_meshlet_instances; /* visibility[0] */
  uint32_t early_visible_meshlet_instances;
  uint32_t late_visible_meshlet_instances;
  uint32_t cull_meshlets_cmd_x;
  uint32_t cull_triangles_cmd_x; /* meshlets emitted by this call */
  uint32_t draw_index_count;     /* 3 * triangles emitted by this call */
} oxc_counters;

/* ---- lifetime ---- */
uint32_t oxc_abi_version(void);
oxc_status oxc_create(int device, oxc_ctx** out);
void oxc_destroy(oxc_ctx* ctx);
const char* oxc_last_error(const oxc_ctx* ctx);

/* Pre-size the context's scratch memory (instance cache, survivor bitmaps, chunk counters) so
 * that no later call allocates.  Optional: calls grow scratch on demand (with a device sync). */
oxc_status oxc_reserve(oxc_ctx* ctx, uint32_t max_mesh_instances, uint32_t max_meshlet_instances);

/* ---- the two reference entry points ---- */
/* Replaces RendererInstance::generate_hiz (Passes/CullGeometry.cpp:10-59, passes/hiz.slang). */
oxc_status oxc_generate_hiz(oxc_ctx* ctx, const oxc_main_geometry_context* context, void* hip_stream);

/* Replaces RendererInstance::cull_geometry (Passes/CullGeometry.cpp:61-404; kernels
 * passes/cull_meshes.slang, cull_meshlets.slang, cull_meshlets_hiz.slang, cull_triangles.slang).
 * Output lists are written in ascending order (a valid outcome of the reference's
 * atomic-ordered output, and a deterministic one) unless context->unordered_output asks for the
 * reference's own atomic slot allocation. */
oxc_status oxc_cull_geometry(oxc_ctx* ctx, const oxc_prepared_frame* frame, oxc_cull_geometry_context* context,
                             void* hip_stream);

/* Makes `hip_stream` wait (on the device) for every triangle stage this context still has in flight on its own stream
 * (calls made with async_triangles = 1).  Cheap when nothing is pending.  While hip_stream is being captured into a HIP graph the
 * context's stream is part of the capture from the first async call on: join before hipStreamEndCapture. */
oxc_status oxc_join_triangles(oxc_ctx* ctx, void* hip_stream);

/* Batched form: semantically `for i < count: oxc_cull_geometry(ctx, &frames[i], &contexts[i], stream)` for
 * INDEPENDENT frames (no buffer of one element is written by another) -- several views or scenes culled per
 * launch, the way the reference's cull_meshlets_hpb handles all clipmap views in one dispatch.  When every
 * element uses the plain pipeline (use_hiz == use_hpb == 0, no LatePass) with the same `stages` and
 * `init_cull_meshes`, and count <= 16, each stage is ONE launch with grid.y = count; otherwise the elements
 * are processed one after the other.  A 1M-meshlet call is launch-latency bound on MI355X (a HIP graph
 * sustains ~3 us per kernel node); batching is what amortises it.
 * When, in addition, every element runs cull_meshes (init_cull_meshes, OXC_STAGE_MESHES) over the SAME meshes_buffer and
 * transforms_world_buffer with the same flags -- several views of one scene: shadow cascades, BASELINE configs[4] -- the meshlet
 * stage runs once for all views: the views that kept a mesh instance at the same LOD are tested against one load of its
 * MeshletBounds records.  Outputs are the per-element outputs of the plain form, byte for byte. */
oxc_status oxc_cull_geometry_batch(oxc_ctx* ctx, uint32_t count, const oxc_prepared_frame* frames,
                                   oxc_cull_geometry_context* contexts, void* hip_stream);

/* Harness helper: start a cull sequence from a caller-provided MeshletInstance list instead of
 * running cull_meshes (fills context->visibility_buffer = {total,0,0} and
 * cull_meshlets_cmd_buffer = {ceil(total/64),1,1}).  The reference always derives these from
 * cull_meshes; the synthetic benchmark configurations start from a given list.
 * Counter buffers handed back in a context (visibility / cull_meshlets_cmd / cull_triangles_cmd /
 * draw_geometry_cmd) are callee-owned slots of a ring: a seeded pair stays valid for 4096 further seeds, a
 * per-call set for 4096 further cull_geometry / cull_terrain calls (a batched call uses one per element; an early call with
 * share_pass_tests takes the late call's set with its own) on the same oxc_ctx. */
oxc_status oxc_seed_meshlet_instances(oxc_ctx* ctx, oxc_cull_geometry_context* context, uint32_t total,
                                      void* hip_stream);

/* Synchronising readback of the counters a context points at (bench / tests). */
oxc_status oxc_read_counters(oxc_ctx* ctx, const oxc_cull_geometry_context* context, oxc_counters* out,
                             void* hip_stream);

/* ---- meshlet bounds producer (asset side) ---------------------------------------
 * Replaces the per-meshlet loop of Oxylus/src/Asset/AssetManager_GLTF.cpp:683-744 (AABB of the referenced
 * vertices -> meshopt_quantizeHalf, normal cone of meshopt_computeMeshletBounds -> cone_axis_s8 /
 * cone_cutoff_s8, running mesh AABB) and the position quantisation of :573-578.  meshopt_buildMeshlets itself
 * (the greedy clusteriser that produces `meshlets`, `indirect_vertex_indices`, `local_triangle_indices`) stays
 * on the host, as in the reference.  All pointers are device pointers. */
typedef struct oxc_meshlet_bounds_desc {
  uint32_t struct_size;
  uint32_t vertex_count;
  uint32_t meshlet_count;
  uint32_t _pad;
  oxc_buffer positions;               /* in:  glm::vec3[vertex_count] (float3, stride 12)             */
  oxc_buffer meshlets;                /* in:  GPU::Meshlet[meshlet_count] (SceneGPU.hpp:97-103)       */
  oxc_buffer indirect_vertex_indices; /* in:  u32, indexed by Meshlet::indirect_vertex_index_offset   */
  oxc_buffer local_triangle_indices;  /* in:  u8,  indexed by Meshlet::local_triangle_index_offset    */
  oxc_buffer meshlet_bounds;          /* out: GPU::MeshletBounds[meshlet_count] (SceneGPU.hpp:84-90)  */
  oxc_buffer mesh_bounds;             /* out: GPU::MeshBounds {vec3 aabb_center; vec3 aabb_extent} (SceneGPU.hpp:92-95) */
  oxc_buffer quantized_positions;     /* out, optional (dptr may be NULL): u16x4[vertex_count]        */
} oxc_meshlet_bounds_desc;

oxc_status oxc_build_meshlet_bounds(oxc_ctx* ctx, const oxc_meshlet_bounds_desc* desc, void* hip_stream);

/* ---- SURVEY 8(f)-1, format side: vertex streams and the mesh blob ---------------------------------
 * oxc_quantize_vertex_streams replaces the three per-vertex loops of AssetManager_GLTF.cpp:570-588:
 *   positions -> u16x4 {half(x), half(y), half(z), 0}                     (:571-575, meshopt_quantizeHalf)
 *   normals   -> u32 ((snorm10(x)+511) << 20) | ((snorm10(y)+511) << 10) | (snorm10(z)+511)
 *                                                                         (:578-582, meshopt_quantizeSnorm(v, 10))
 *   texcoords -> u16x2 {half(u), half(v)}                                 (:585-588)
 * A stream whose input dptr is NULL is skipped (its output is not touched).  Device pointers. */
typedef struct oxc_vertex_streams_desc {
  uint32_t struct_size;
  uint32_t vertex_count;
  oxc_buffer positions;           /* in, optional:  glm::vec3[vertex_count] */
  oxc_buffer normals;             /* in, optional:  glm::vec3[vertex_count] */
  oxc_buffer texcoords;           /* in, optional:  glm::vec2[vertex_count] */
  oxc_buffer quantized_positions; /* out: u16x4[vertex_count] */
  oxc_buffer quantized_normals;   /* out: u32[vertex_count]   */
  oxc_buffer quantized_texcoords; /* out: u16x2[vertex_count] */
} oxc_vertex_streams_desc;

oxc_status oxc_quantize_vertex_streams(oxc_ctx* ctx, const oxc_vertex_streams_desc* desc, void* hip_stream);

/* The mesh blob: one allocation per mesh holding the vertex streams, every LOD's five arrays and, last, the
 * GPU::MeshLOD table (AssetManager_GLTF.cpp:466-474 blob_append, :590-597, :748-752, :768-769).  Offsets follow
 * blob_append's rule offset = align_up(current size, alignment): positions 8, normals 4, texcoords 4 (only when
 * present), then per LOD indices 8, meshlets 8, meshlet_bounds 8, local_triangle_indices 8,
 * indirect_vertex_indices 4, then the LOD table at align_up(size, 8).  Host-only arithmetic: no context, no GPU. */
#define OXC_MESH_MAX_LODS 8u /* GPU::Mesh::MAX_LODS, SceneGPU.hpp:143 */

typedef struct oxc_mesh_lod_counts { /* the *_count fields of GPU::MeshLOD (SceneGPU.hpp:125-139) + error */
  uint32_t indices_count;                 /* u32 elements */
  uint32_t meshlet_count;                 /* GPU::Meshlet (16 B) and GPU::MeshletBounds (16 B) records */
  uint32_t local_triangle_indices_count;  /* u8 elements (last meshlet's run padded to 4, :699) */
  uint32_t indirect_vertex_indices_count; /* u32 elements */
  float error;
} oxc_mesh_lod_counts;

typedef struct oxc_mesh_blob_desc {
  uint32_t struct_size;
  uint32_t vertex_count;
  uint32_t has_texture_coords;
  uint32_t lod_count; /* 1..OXC_MESH_MAX_LODS */
  oxc_mesh_lod_counts lods[OXC_MESH_MAX_LODS];
} oxc_mesh_blob_desc;

typedef struct oxc_mesh_lod_offsets {
  uint64_t indices, meshlets, meshlet_bounds, local_triangle_indices, indirect_vertex_indices;
} oxc_mesh_lod_offsets;

typedef struct oxc_mesh_blob_layout { /* byte offsets from the start of the blob */
  uint64_t size;                /* whole blob, LOD table included */
  uint64_t lod_metadata_offset; /* GPU::MeshLOD[lod_count] */
  uint64_t vertex_positions, vertex_normals, texture_coords; /* texture_coords = 0 when absent */
  oxc_mesh_lod_offsets lods[OXC_MESH_MAX_LODS];
} oxc_mesh_blob_layout;

oxc_status oxc_mesh_blob_layout_of(const oxc_mesh_blob_desc* desc, oxc_mesh_blob_layout* out_layout);

/* upload_gltf_mesh's relocation (AssetManager_GLTF.cpp:780-800): with the blob resident at `device_address`,
 * write the GPU::MeshLOD table (absolute addresses + counts + error) into the HOST copy `blob` at
 * lod_metadata_offset and fill `out_gpu_mesh` (64 B GPU::Mesh: absolute stream addresses, texture_coords 0 when
 * absent, vertex_count, lod_count, lods, bounds = mesh_bounds {aabb_center.xyz, aabb_extent.xyz}).  The caller
 * then copies the blob to the device (the reference's staging upload, :802-818). */
oxc_status oxc_mesh_blob_finalize(const oxc_mesh_blob_desc* desc, const oxc_mesh_blob_layout* layout, uint64_t device_address,
                                  void* blob, uint64_t blob_bytes, const float mesh_bounds[6], void* out_gpu_mesh);

/* ---- SURVEY 8(f)-1, clusteriser side: triangle soup -> LOD chain -> meshlets (host code, as in the reference) ----------
 * Replaces the per-LOD loop of AssetManager_GLTF.cpp:599-682: LOD 0 = the input indices, LOD i = LOD i-1 simplified to half its
 * index count with the border locked (meshopt_simplifyWithAttributes, normal weights 1), error accumulated down the chain, chain
 * cut by the reference's three stop rules (:639-645) or at GPU::Mesh::MAX_LODS; every LOD clustered into meshlets of at most
 * max_vertices / max_triangles (64 / 64: Model::MAX_MESHLET_INDICES / _PRIMITIVES, meshopt_buildMeshlets with cone_weight 0), u8
 * micro-index runs 4-byte aligned (:687).  meshoptimizer is a third-party dependency that is not part of the reference tree: its two
 * algorithms are restated in shape, not heuristic for heuristic (oxylus_amd/csrc/oxcull_meshbuild.cpp) -- a valid, different clustering.
 * LOD 0's `indices` are the input verbatim; triangles with a repeated corner are left out of the meshlets and of the simplifier's input
 * (no area).  `error` accumulates this simplifier's own relative error measure: it orders a mesh's LODs like meshopt's result_error
 * does but is not numerically comparable to it, so CULL_SELECT_LOD thresholds tuned against meshoptimizer do not carry over.
 * Host pointers in, host views out (owned by the handle); feed them to oxc_build_meshlet_bounds / oxc_quantize_vertex_streams /
 * oxc_mesh_blob_* to obtain what oxc_cull_geometry consumes. */
typedef struct oxc_mesh_build oxc_mesh_build;
typedef struct oxc_mesh_build_desc {
  uint32_t struct_size;
  uint32_t vertex_count;
  uint32_t index_count;   /* multiple of 3 */
  uint32_t max_lods;      /* 0 = OXC_MESH_MAX_LODS */
  uint32_t max_vertices;  /* per meshlet, 0 = 64 */
  uint32_t max_triangles; /* per meshlet, 0 = 64 */
  const float* positions; /* glm::vec3[vertex_count] */
  const float* normals;   /* glm::vec3[vertex_count], optional (NULL: positions only) */
  const uint32_t* indices;
} oxc_mesh_build_desc;
/* (LOD 0 of a mesh whose input has degenerate triangles: `indices` / `indices_count` keep them -- the reference passes the index buffer
 * through unchanged -- while the meshlets are built from the non-degenerate ones, so indices_count / 3 can exceed the sum of the meshlets'
 * triangle_count.  Nothing on the cull path reads `indices`; a consumer that draws from them gets the degenerate triangles back, which
 * rasterise nothing.) */
typedef struct oxc_mesh_lod_view { /* the arrays of one GPU::MeshLOD (SceneGPU.hpp:125-139), host memory */
  const uint32_t* indices;
  const void* meshlets; /* GPU::Meshlet[meshlet_count], 16 B each */
  const uint32_t* indirect_vertex_indices;
  const uint8_t* local_triangle_indices;
  uint32_t indices_count, meshlet_count, indirect_vertex_indices_count, local_triangle_indices_count;
  float error;
  uint32_t _pad;
} oxc_mesh_lod_view;
oxc_status oxc_mesh_build_create(const oxc_mesh_build_desc* desc, oxc_mesh_build** out);
uint32_t oxc_mesh_build_lod_count(const oxc_mesh_build* build);
oxc_status oxc_mesh_build_lod(const oxc_mesh_build* build, uint32_t lod, oxc_mesh_lod_view* out);
void oxc_mesh_build_destroy(oxc_mesh_build* build);
/* Replaces meshopt_optimizeVertexFetchRemap as AssetManager_GLTF.cpp:512-568 uses it (host code there too): remap_out[old vertex id] = its
 * rank by first appearance in `stream` (count entries, each < vertex_count); vertices the stream never names follow the used ones in their
 * old order (the reference drops them: *used_out tells how many are used).  The reference passes the raw index buffer, before it simplifies
 * and clusters.  Passing LOD 0's indirect_vertex_indices instead orders the vertices by MESHLET: the <= 64 vertices of a meshlet then lie
 * next to each other in vertex_positions and cull_triangles' position gather touches a handful of cache lines (measured: DESIGN.md 5).  The
 * caller applies the remap to its vertex streams and to every LOD's indices / indirect_vertex_indices; geometry and meshlets do not change. */
oxc_status oxc_mesh_vertex_fetch_remap(const uint32_t* stream, uint64_t count, uint32_t vertex_count, uint32_t* remap_out, uint32_t* used_out);

/* ---- hierarchical page buffer producer ------------------------------------------
 * Replaces the "vsm downsample hpb" pass (Oxylus/src/Render/Passes/Shadowmaps.cpp:331-366, pipeline
 * rmvsm_downsample_hpb, Shaders/passes/rmvsm_downsample_hpb.slang:10-33): level 0 of the pyramid is 1 where
 * the virtual page is Visible && Backed && Dirty (VSMPageState flags 1, 4, 2: rmvsm.slang:16-28,49-70), level i
 * is the OR of the 2x2 children in level i-1 (texels outside the source level read as 0).
 * `virtual_page_table` is the R32UI Texture2DArray as a linear u32 array [layers][height][width];
 * the extent of level i is max(1, width >> i) x max(1, height >> i) (Shadowmaps.cpp:342-346). */
oxc_status oxc_generate_hpb(oxc_ctx* ctx, oxc_buffer virtual_page_table, const oxc_image_array_u8* hpb_attachment, void* hip_stream);

/* ---- terrain patch cull ---------------------------------------------------------
 * Replaces RendererInstance::cull_terrain (Oxylus/src/Render/Passes/Terrain.cpp:159-216) + pipeline
 * terrain_cull (Shaders/passes/terrain_cull.slang:17-83): one thread per patch, world-space AABB from the
 * patch grid and the patch_minmax image, the same test_frustum / project_aabb / test_occlusion and early/late
 * mask protocol as cull_meshlets_hiz, survivors appended to visible_patches and counted in
 * DrawIndirectCommand.instance_count.  The reference appends in atomic order; here the list is ascending. */
typedef struct oxc_terrain_context {
  uint32_t struct_size;
  uint32_t cull_flags;               /* OXC_CULL_* (TestFrustum, TestOcclusion, LatePass) */
  oxc_cull_camera cull_camera;       /* projection_view, near_clip; mesh_instance_count is set by the callee (Terrain.cpp:171) */
  /* the GPU::TerrainData fields the shader reads (SceneGPU.hpp:440-453, scene.slang:634-661) */
  float world_min[2];
  float world_size[2];
  uint32_t patch_count[2];
  float base_height;
  float height_scale;
  oxc_image patch_minmax_attachment; /* RG32F, patch_count.x x patch_count.y, levels = 1: {min, max} normalised height per patch */
  oxc_image hiz_attachment;          /* read when TestOcclusion or LatePass */
  oxc_buffer visible_patches_buffer; /* out: u32[patch_total] */
  oxc_buffer patch_visibility_mask_buffer; /* in/out: u32[ceil(patch_total / 32)] */
  oxc_buffer draw_cmd_buffer;        /* out (callee-owned, like the reference's scratch_buffer): VkDrawIndirectCommand {4, instance_count, 0, 0} */
} oxc_terrain_context;

oxc_status oxc_cull_terrain(oxc_ctx* ctx, oxc_terrain_context* context, void* hip_stream);

/* ---- consumer of the indirect draw ------------------------------------------------
 * What RendererInstance::draw_for_visbuffer does with cull_geometry's outputs (Passes/DrawGeometry.cpp:104-190,
 * pipeline visbuffer_encode, passes/visbuffer_encode.slang:24-49; cullMode eBack, depth GreaterOrEqual,
 * reversed Z), as a compute rasteriser -- so that early cull -> draw -> depth -> generate_hiz -> late cull -> draw
 * runs without a graphics queue.  The fixed-function rasteriser's sample rules cannot be matched bit for bit;
 * the rules used instead are stated here and in the checker:
 *   vertex  VisBufferData(index) -> (meshlet instance, corner); Meshlet::index, Mesh::decode_position;
 *           world = mul(world, (p,1)).xyz, clip = mul(projection_view, (world,1))              (as vs_main);
 *   clip    Sutherland-Hodgman against five planes in this order: w >= 2^-10, 64 w - x >= 0, 64 w + x >= 0, 64 w - y >= 0,
 *           64 w + y >= 0 (a 64x guard band: every screen coordinate stays inside the +-2^20 px fixed-point range for extents up
 *           to 16384; the reference's fixed-function clipper stands here).  A vertex with distance d >= 0 is inside.  On a crossing
 *           edge with inside end I and outside end O the new vertex is I + t (O - I), t = d(I) / (d(I) - d(O)), binary32, no
 *           contraction, all four clip coordinates -- the same value for both triangles that share the edge.  The polygon
 *           (<= 8 corners) is drawn as the fan (p0, pk, pk+1), each with the vis value of the source triangle; a triangle inside
 *           every plane is untouched (round 1 had no clipper and dropped triangles with a corner at w <= 0).  Capacity: the
 *           ids of the triangles that cross a plane are queued, 2^22 per call; when more cross, an overflow pass walks the index
 *           list again and clips every crossing triangle it finds (slow, and the same image: drawing a triangle twice changes nothing);
 *   setup   screen = (clip.xy / clip.w * 0.5 + 0.5) * extent, snapped to 1/256 pixel; back faces (fixed-point
 *           area >= 0, the orientation cull_triangles' determinant test calls back-facing) are dropped;
 *   cover   pixel centres, integer edge functions, top-left rule;
 *   depth   z/w interpolated in binary64, ((e0 z0 + e1 z1) + e2 z2) * (1 / area) with the exact integer edge values e_i, rounded to
 *           binary32, kept when in (0, 1];
 *           per pixel the maximum of (depth bits << 32) | vis wins (64-bit atomic max), vis = (instance << 8) |
 *           (corner / 3) as VisBufferData::encode -- order-independent: one of the results the reference's
 *           equal-depth race can produce.
 * `visdepth_buffer` (u64[width * height]) persists between the early and the late draw of a frame; `clear` zeroes
 * it first.  `depth_attachment` (R32F, levels = 1) and `visbuffer_attachment` (u32) are optional resolves. */
typedef struct oxc_draw_context {
  uint32_t struct_size;
  uint32_t wide_triangle_index; /* same meaning as in oxc_cull_geometry_context */
  uint32_t clear;
  uint32_t width, height;
  uint32_t _pad;
  float projection_view[16];           /* Camera::projection_view, column-major */
  oxc_buffer draw_geometry_cmd_buffer; /* from oxc_cull_geometry: indexCount is read on the device */
  oxc_buffer visdepth_buffer;
  oxc_image depth_attachment;          /* optional out */
  oxc_buffer visbuffer_attachment;     /* optional out */
} oxc_draw_context;

oxc_status oxc_draw_visbuffer(oxc_ctx* ctx, const oxc_prepared_frame* frame, const oxc_draw_context* context, void* hip_stream);

/* ---- multi-GPU exchange: one process per GPU, RCCL over xGMI ---------------------------
 * The meshlet-instance array shards by contiguous range and every rank culls its shard on its own; the only
 * exchanges of the path are (1) the per-rank counters {emitted meshlets, early, late, index count} to every rank
 * (16 bytes per rank) so that each can place its compacted buffers in a merged list, and (2) the HiZ pyramid from
 * the rank that owns the depth buffer.  These entry points are those two collectives on the caller's stream,
 * through RCCL (loaded with dlopen, so the single-GPU path does not need the library).
 *   rank 0: oxc_comm_unique_id(id) -> the launcher hands the 128 bytes to the other ranks (any side channel)
 *   all:    oxc_comm_init(ctx, id, rank, world)                                                          */
#define OXC_COMM_UNIQUE_ID_BYTES 128
oxc_status oxc_comm_unique_id(oxc_ctx* ctx, void* id128_host_out);
oxc_status oxc_comm_init(oxc_ctx* ctx, const void* id128_host, uint32_t rank, uint32_t world);
oxc_status oxc_comm_destroy(oxc_ctx* ctx);
/* Packs the counters of `context`'s last oxc_cull_geometry call -- {meshlets emitted (cull_triangles_cmd.x), early, late
 * (visibility_buffer), index_count (draw_geometry_cmd)} -- into counts4_dptr (device, u32[4]) on the stream: the input of
 * oxc_exchange_counts, without a host round trip.  Usable on one GPU as well. */
oxc_status oxc_pack_counters(oxc_ctx* ctx, const oxc_cull_geometry_context* context, void* counts4_dptr, void* hip_stream);
/* The same for the `count` (<= 16) contexts of one oxc_cull_geometry_batch call in ONE launch: counts4_dptr[i][4] = element i's
 * {emitted, visibility.total (the length of the view's MeshletInstance list), late, index_count} -- configs[4] sharded over ranks all-gathers
 * these per view. */
oxc_status oxc_pack_counters_batch(oxc_ctx* ctx, uint32_t count, const oxc_cull_geometry_context* contexts, void* counts4_dptr, void* hip_stream);
/* all-gather of 4 u32 per rank: counts4_dptr (this rank's {emitted, early, late, index_count}) -> all_counts_dptr[world][4] */
oxc_status oxc_exchange_counts(oxc_ctx* ctx, const void* counts4_dptr, void* all_counts_dptr, void* hip_stream);
/* broadcast of every level of `hiz` from rank `root` (in place) */
oxc_status oxc_broadcast_hiz(oxc_ctx* ctx, const oxc_image* hiz, uint64_t total_bytes, uint32_t root, void* hip_stream);
/* The "top mips" form of the same exchange: only levels >= first_level travel (bytes [level_offset[first_level], total_bytes) of the
 * linear chain: 5.6 MB instead of 89.5 MB for a 4096^2 pyramid with first_level = 2).  A rank must hold EVERY level it may sample
 * (clamping the mip would change results), so this form is for ranks that own a copy of the depth image and build levels
 * < first_level themselves: oxc_generate_hiz with hiz_attachment.levels = first_level.  The pyramid is a pure function of the depth
 * image, so both forms give every rank the same bytes. */
oxc_status oxc_broadcast_hiz_levels(oxc_ctx* ctx, const oxc_image* hiz, uint32_t first_level, uint64_t total_bytes, uint32_t root, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* OXCULL_H */
