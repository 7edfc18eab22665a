// RendererInstance.hpp -- C++ drop-in shim over the C ABI (include/oxcull.h).
//
// Keeps the reference's names and field meaning for the cull path so that the engine-side
// code of RendererInstance::render (Oxylus/src/Render/RendererInstance.cpp:793-884) compiles
// against it with `vuk::Value<vuk::Buffer>` replaced by `ox::amd::Buffer` (device pointer +
// size) and `vuk::Value<vuk::ImageAttachment>` by `ox::amd::ImageAttachment` (linear mip
// chain).  Mirrors:
//   GPU::CullFlag / GPU::CullCamera           Oxylus/include/Scene/SceneGPU.hpp:222-229,345-353
//   PreparedFrame                             Oxylus/include/Render/RendererInstance.hpp:143-169
//   CullGeometryContext / MainGeometryContext Oxylus/include/Render/RendererInstance.hpp:171-216
//   RendererInstance::generate_hiz / cull_geometry            ...RendererInstance.hpp:397-398
// Error behaviour: the reference's entry points return void and abort through OX_CHECK_* on
// programmer errors (Oxylus/include/Utils/Log.hpp:38-47); the shim throws std::runtime_error
// carrying oxc_last_error() instead of aborting.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>

#include "oxcull.h"

namespace ox::amd {

namespace GPU {
enum struct CullFlag : uint32_t {
  None = 0,
  TestFrustum = 1 << 0,
  SelectLOD = 1 << 1,
  TestOcclusion = 1 << 2,
  LatePass = 1 << 3,
  TestAll = TestFrustum | SelectLOD | TestOcclusion,
};
constexpr CullFlag operator|(CullFlag a, CullFlag b) { return static_cast<CullFlag>(static_cast<uint32_t>(a) | static_cast<uint32_t>(b)); }
constexpr CullFlag& operator|=(CullFlag& a, CullFlag b) { return a = a | b; }
constexpr bool operator&(CullFlag a, CullFlag b) { return (static_cast<uint32_t>(a) & static_cast<uint32_t>(b)) != 0; }

using CullCamera = oxc_cull_camera;  // same 96-byte layout as GPU::CullCamera
static_assert(sizeof(CullCamera) == 96, "GPU::CullCamera is a 96-byte push constant");
}  // namespace GPU

using Buffer = oxc_buffer;             // stands in for vuk::Value<vuk::Buffer>
using ImageAttachment = oxc_image;     // stands in for vuk::Value<vuk::ImageAttachment> (R32F / D32F)
using ImageArrayAttachment = oxc_image_array_u8;  // the R8UI mip-mapped array `hpb_attachment`

struct PreparedFrame {
  uint32_t mesh_instance_count = 0;
  uint32_t max_meshlet_instance_count = 0;
  bool use_mesh_shaders = false;  // the mesh-shader path needs graphics hardware: must stay false
  Buffer transforms_world_buffer = {};
  Buffer meshes_buffer = {};
  Buffer mesh_instances_buffer = {};
  Buffer meshlet_instances_buffer = {};
  Buffer visible_meshlet_instances_indices_buffer = {};
  Buffer meshlet_instance_visibility_mask_buffer = {};
  Buffer reordered_indices_buffer = {};
};

struct CullGeometryContext {
  bool use_hiz = false;
  bool use_hpb = false;
  bool init_cull_meshes = false;
  GPU::CullFlag cull_flags = GPU::CullFlag::TestAll;
  GPU::CullCamera cull_camera = {};
  Buffer vsm_clipmaps_buffer = {};
  Buffer vsm_clipmap_dirty_flags_buffer = {};
  uint32_t vsm_clipmap_count = 0;
  ImageAttachment hiz_attachment = {};
  ImageArrayAttachment hpb_attachment = {};
  Buffer visibility_buffer = {};
  Buffer cull_meshlets_cmd_buffer = {};
  Buffer draw_geometry_cmd_buffer = {};
  // not in the reference struct (it is a local there, CullGeometry.cpp:125-127): the indirect
  // dispatch command of cull_triangles, exposed so callers can read the visible-meshlet count
  Buffer cull_triangles_cmd_buffer = {};
  // extension: 1 = wide packed index for meshlets of up to 128 triangles (<= 2^23 ids), 2 = {id, corner} pairs (no id limit)
  uint32_t wide_triangle_index = 0;
  bool small_triangle_cull = false;  // extension named by the north star; default OFF = reference behaviour
  // extension (scheduling only): cull_triangles of this call runs on the backend's own stream beside what the caller enqueues next
  // (the next cull_geometry's meshlet stage, generate_hiz); join_triangles() before anything of the caller's reads the index list
  bool async_triangles = false;
  // extension (caching only): set on both HiZ calls of a frame (early, then LatePass with the same camera and buffers), the late call
  // takes the frustum + cone results from the early one instead of testing again -- the caller vouches that no input of those tests
  // was written in between (include/oxcull.h: share_pass_tests)
  bool share_pass_tests = false;
  // extension (order only): 0 = ascending lists; 1 / 2 = the reference's own atomic slot allocation, aggregated per block / wave step
  // (one launch per stage instead of test + ordered emit; include/oxcull.h: unordered_output)
  uint32_t unordered_output = 0;
  // extension (oxc_cull_geometry_batch's multi-view path only): cull_meshes leaves the MeshletInstance list implicit and writes
  // {first, count} per mesh instance here instead (include/oxcull.h: implicit_meshlet_instances)
  bool implicit_meshlet_instances = false;
  Buffer meshlet_instance_runs_buffer = {};
};

struct MainGeometryContext {
  GPU::CullFlag cull_flags = GPU::CullFlag::TestAll;
  GPU::CullCamera cull_camera = {};
  ImageAttachment depth_attachment = {};
  ImageAttachment hiz_attachment = {};
  Buffer visbuffer_attachment = {};  // R32_UINT image as a linear buffer of width * height u32 (the reference: vuk::ImageAttachment)
  Buffer draw_geometry_cmd_buffer = {};
  Buffer visibility_buffer = {};
  // not in the reference struct: the compute rasteriser's packed depth|vis image (u64 per pixel) that persists between the early
  // and the late draw of a frame, whether this draw starts from a cleared image (the early one), and the wide-index extension
  Buffer visdepth_buffer = {};
  bool clear = true;
  uint32_t wide_triangle_index = 0;
};

class RendererInstance {
public:
  explicit RendererInstance(int device = 0, void* hip_stream = nullptr) : stream_(hip_stream) {
    if (oxc_create(device, &ctx_) != OXC_OK) throw std::runtime_error("oxc_create failed");
  }
  ~RendererInstance() { oxc_destroy(ctx_); }
  RendererInstance(const RendererInstance&) = delete;
  RendererInstance& operator=(const RendererInstance&) = delete;

  PreparedFrame prepared_frame = {};

  void set_stream(void* hip_stream) { stream_ = hip_stream; }

  // Oxylus/src/Render/Passes/CullGeometry.cpp:10-59
  auto generate_hiz(MainGeometryContext& context) -> void {
    oxc_main_geometry_context c = {};
    c.struct_size = sizeof c;
    c.depth_attachment = context.depth_attachment;
    c.hiz_attachment = context.hiz_attachment;
    check(oxc_generate_hiz(ctx_, &c, stream_));
  }

  // Oxylus/src/Render/Passes/CullGeometry.cpp:61-404
  auto cull_geometry(CullGeometryContext& context) -> void {
    if (prepared_frame.use_mesh_shaders) throw std::runtime_error("cull_geometry: the mesh-shader path is not available on the compute-only backend");
    oxc_prepared_frame f = {};
    f.mesh_instance_count = prepared_frame.mesh_instance_count;
    f.max_meshlet_instance_count = prepared_frame.max_meshlet_instance_count;
    f.meshes_buffer = prepared_frame.meshes_buffer;
    f.transforms_world_buffer = prepared_frame.transforms_world_buffer;
    f.mesh_instances_buffer = prepared_frame.mesh_instances_buffer;
    f.meshlet_instances_buffer = prepared_frame.meshlet_instances_buffer;
    f.visible_meshlet_instances_indices_buffer = prepared_frame.visible_meshlet_instances_indices_buffer;
    f.meshlet_instance_visibility_mask_buffer = prepared_frame.meshlet_instance_visibility_mask_buffer;
    f.reordered_indices_buffer = prepared_frame.reordered_indices_buffer;
    oxc_cull_geometry_context c = {};
    c.struct_size = sizeof c;
    c.use_hiz = context.use_hiz;
    c.use_hpb = context.use_hpb;
    c.init_cull_meshes = context.init_cull_meshes;
    c.cull_flags = static_cast<uint32_t>(context.cull_flags);
    c.cull_camera = context.cull_camera;
    c.hiz_attachment = context.hiz_attachment;
    c.hpb_attachment = context.hpb_attachment;
    c.vsm_clipmaps_buffer = context.vsm_clipmaps_buffer;
    c.vsm_clipmap_dirty_flags_buffer = context.vsm_clipmap_dirty_flags_buffer;
    c.vsm_clipmap_count = context.vsm_clipmap_count;
    c.wide_triangle_index = context.wide_triangle_index;
    c.small_triangle_cull = context.small_triangle_cull;
    c.async_triangles = context.async_triangles;
    c.share_pass_tests = context.share_pass_tests;
    c.unordered_output = context.unordered_output;
    c.implicit_meshlet_instances = context.implicit_meshlet_instances;
    c.meshlet_instance_runs_buffer = context.meshlet_instance_runs_buffer;
    c.visibility_buffer = context.visibility_buffer;
    c.cull_meshlets_cmd_buffer = context.cull_meshlets_cmd_buffer;
    check(oxc_cull_geometry(ctx_, &f, &c, stream_));
    context.visibility_buffer = c.visibility_buffer;
    context.cull_meshlets_cmd_buffer = c.cull_meshlets_cmd_buffer;
    context.cull_triangles_cmd_buffer = c.cull_triangles_cmd_buffer;
    context.draw_geometry_cmd_buffer = c.draw_geometry_cmd_buffer;
  }

  // The stream waits for every cull_triangles stage still in flight (async_triangles); draw_for_visbuffer joins by itself.
  auto join_triangles() -> void { check(oxc_join_triangles(ctx_, stream_)); }

  // Oxylus/src/Render/Passes/DrawGeometry.cpp:104-190 for the compute-only backend: the triangles of context.draw_geometry_cmd_buffer
  // (prepared_frame.reordered_indices_buffer, as vs_main decodes them) into depth_attachment / visbuffer_attachment, with
  // context.cull_camera.projection_view as Camera::projection_view.  Rules: include/oxcull.h, oxc_draw_visbuffer.
  auto draw_for_visbuffer(MainGeometryContext& context) -> void {
    if (prepared_frame.use_mesh_shaders) throw std::runtime_error("draw_for_visbuffer: the mesh-shader path is not available on the compute-only backend");
    oxc_prepared_frame f = {};
    f.mesh_instance_count = prepared_frame.mesh_instance_count;
    f.max_meshlet_instance_count = prepared_frame.max_meshlet_instance_count;
    f.meshes_buffer = prepared_frame.meshes_buffer;
    f.transforms_world_buffer = prepared_frame.transforms_world_buffer;
    f.mesh_instances_buffer = prepared_frame.mesh_instances_buffer;
    f.meshlet_instances_buffer = prepared_frame.meshlet_instances_buffer;
    f.visible_meshlet_instances_indices_buffer = prepared_frame.visible_meshlet_instances_indices_buffer;
    f.meshlet_instance_visibility_mask_buffer = prepared_frame.meshlet_instance_visibility_mask_buffer;
    f.reordered_indices_buffer = prepared_frame.reordered_indices_buffer;
    oxc_draw_context d = {};
    d.struct_size = sizeof d;
    d.wide_triangle_index = context.wide_triangle_index;
    d.clear = context.clear;
    d.width = context.depth_attachment.width;
    d.height = context.depth_attachment.height;
    for (int i = 0; i < 16; i++) d.projection_view[i] = context.cull_camera.projection_view[i];
    d.draw_geometry_cmd_buffer = context.draw_geometry_cmd_buffer;
    d.visdepth_buffer = context.visdepth_buffer;
    d.depth_attachment = context.depth_attachment;
    d.visbuffer_attachment = context.visbuffer_attachment;
    check(oxc_draw_visbuffer(ctx_, &f, &d, stream_));
  }

  // Oxylus/src/Render/Passes/Terrain.cpp:159-216 (the reference's TerrainContext carries the Terrain object; here its GPU-side fields)
  auto cull_terrain(oxc_terrain_context& context) -> void {
    context.struct_size = sizeof context;
    check(oxc_cull_terrain(ctx_, &context, stream_));
  }

  // Producers around the cull path: the downsample_hpb_pass of draw_virtual_shadowmap
  // (Passes/Shadowmaps.cpp:331-366) and the per-meshlet bounds loop of the asset import
  // (Asset/AssetManager_GLTF.cpp:573-578,683-744).
  auto generate_hpb(oxc_buffer virtual_page_table, const oxc_image_array_u8& hpb_attachment) -> void {
    check(oxc_generate_hpb(ctx_, virtual_page_table, &hpb_attachment, stream_));
  }
  auto build_meshlet_bounds(oxc_meshlet_bounds_desc desc) -> void {
    desc.struct_size = sizeof desc;
    check(oxc_build_meshlet_bounds(ctx_, &desc, stream_));
  }
  // AssetManager_GLTF.cpp:570-588: the three quantised vertex streams of the mesh blob
  auto quantize_vertex_streams(oxc_vertex_streams_desc desc) -> void {
    desc.struct_size = sizeof desc;
    check(oxc_quantize_vertex_streams(ctx_, &desc, stream_));
  }

  oxc_ctx* native() { return ctx_; }

private:
  void check(oxc_status st) {
    if (st != OXC_OK) throw std::runtime_error(std::string("oxcull: ") + oxc_last_error(ctx_));
  }
  oxc_ctx* ctx_ = nullptr;
  void* stream_ = nullptr;
};

}  // namespace ox::amd
