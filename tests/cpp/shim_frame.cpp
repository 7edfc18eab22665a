// tests/cpp/shim_frame.cpp -- drives ox::amd::RendererInstance (oxylus_amd/host/RendererInstance.hpp, the C++ surface an engine
// would include) with REAL buffers, in the order of RendererInstance::render's 3D pass (Oxylus/src/Render/RendererInstance.cpp:
// 793-884), and dumps every output for tests/test_cpp_shim.py to compare with the checker / the committed fixture.
//
//   shim_frame <in.oxcf> <out.oxcf>
//
// Sequence A (plain pipeline, CullGeometry.cpp:61-404 with use_hiz = false):
//     cull_geometry({init_cull_meshes = true, cull_flags = TestAll})
// Sequence B (two-pass occlusion, RendererInstance.cpp:842-884):
//     cull_geometry({use_hiz, init_cull_meshes = true, TestAll, hiz = last frame's pyramid})          -- early
//     [the engine draws the early list here; the test supplies the resulting depth image instead]
//     generate_hiz({depth_attachment = depth1, hiz_attachment})
//     cull_geometry({use_hiz, init_cull_meshes = false, TestAll | LatePass, same hoisted context})    -- late
// Container format (little endian): "OXCF", u32 count, then count x {char name[24]; u64 offset; u64 bytes}, then the payloads.
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "RendererInstance.hpp"

using namespace ox::amd;

namespace {
struct Entry {
  char name[24];
  uint64_t offset, bytes;
};
using Blob = std::vector<uint8_t>;

std::map<std::string, Blob> read_container(const char* path) {
  FILE* f = std::fopen(path, "rb");
  if (!f) throw std::runtime_error(std::string("cannot open ") + path);
  std::fseek(f, 0, SEEK_END);
  long size = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  Blob all((size_t)size);
  if (std::fread(all.data(), 1, all.size(), f) != all.size()) throw std::runtime_error("short read");
  std::fclose(f);
  if (std::memcmp(all.data(), "OXCF", 4) != 0) throw std::runtime_error("bad magic");
  uint32_t n;
  std::memcpy(&n, all.data() + 4, 4);
  std::map<std::string, Blob> out;
  for (uint32_t i = 0; i < n; i++) {
    Entry e;
    std::memcpy(&e, all.data() + 8 + i * sizeof(Entry), sizeof e);
    out[std::string(e.name)] = Blob(all.begin() + (long)e.offset, all.begin() + (long)(e.offset + e.bytes));
  }
  return out;
}

void write_container(const char* path, const std::vector<std::pair<std::string, Blob>>& items) {
  FILE* f = std::fopen(path, "wb");
  if (!f) throw std::runtime_error(std::string("cannot write ") + path);
  uint32_t n = (uint32_t)items.size();
  std::fwrite("OXCF", 1, 4, f);
  std::fwrite(&n, 4, 1, f);
  uint64_t off = 8 + (uint64_t)n * sizeof(Entry);
  for (auto& it : items) {
    Entry e = {};
    std::strncpy(e.name, it.first.c_str(), sizeof e.name - 1);
    e.offset = off;
    e.bytes = it.second.size();
    std::fwrite(&e, sizeof e, 1, f);
    off += e.bytes;
  }
  for (auto& it : items) std::fwrite(it.second.data(), 1, it.second.size(), f);
  std::fclose(f);
}

#define HIP(x)                                                                                     \
  do {                                                                                             \
    hipError_t e_ = (x);                                                                           \
    if (e_ != hipSuccess) throw std::runtime_error(std::string(#x) + ": " + hipGetErrorString(e_)); \
  } while (0)

struct DeviceBuffers {
  std::map<std::string, void*> ptr;
  std::map<std::string, uint64_t> bytes;
  void* alloc(const std::string& name, uint64_t n, int fill = 0) {
    void* p = nullptr;
    HIP(hipMalloc(&p, n ? n : 4));
    HIP(hipMemset(p, fill, n ? n : 4));
    ptr[name] = p;
    bytes[name] = n;
    return p;
  }
  Buffer buf(const std::string& name) { return Buffer{ptr.at(name), bytes.at(name)}; }
  Blob download(const std::string& name, uint64_t n) {
    Blob b(n);
    if (n) HIP(hipMemcpy(b.data(), ptr.at(name), n, hipMemcpyDeviceToHost));
    return b;
  }
  ~DeviceBuffers() {
    for (auto& kv : ptr) (void)hipFree(kv.second);
  }
};

// relocation record: the u64 at `field_offset` of section `field_section` = device address of `target_section` + addend
struct Reloc {
  char field_section[24], target_section[24];
  uint64_t field_offset, addend;
};

struct HizDesc {
  uint32_t width, height, levels, _pad;
  uint64_t level_offset[13];
  uint64_t total_bytes;
};

ImageAttachment image_of(void* dptr, const HizDesc& d) {
  ImageAttachment im = {};
  im.dptr = dptr;
  im.width = d.width;
  im.height = d.height;
  im.levels = d.levels;
  for (int k = 0; k < 13; k++) im.level_offset[k] = d.level_offset[k];
  return im;
}

Blob counters(RendererInstance& self, const CullGeometryContext& c) {
  oxc_cull_geometry_context cc = {};
  cc.struct_size = sizeof cc;
  cc.visibility_buffer = c.visibility_buffer;
  cc.cull_meshlets_cmd_buffer = c.cull_meshlets_cmd_buffer;
  cc.cull_triangles_cmd_buffer = c.cull_triangles_cmd_buffer;
  cc.draw_geometry_cmd_buffer = c.draw_geometry_cmd_buffer;
  oxc_counters out = {};
  if (oxc_read_counters(self.native(), &cc, &out, nullptr) != OXC_OK) throw std::runtime_error(oxc_last_error(self.native()));
  Blob b(sizeof out);
  std::memcpy(b.data(), &out, sizeof out);
  return b;
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) {
    std::puts("usage: shim_frame <in.oxcf> <out.oxcf>");
    return 2;
  }
  try {
    auto in = read_container(argv[1]);
    HIP(hipSetDevice(0));
    DeviceBuffers dev;
    const char* scene_sections[] = {"bounds", "meshlets", "micro", "vidx", "positions", "lods", "meshes", "transforms", "mesh_instances", "depth0", "depth1", "mask_in"};
    for (const char* s : scene_sections) dev.alloc(s, in.at(s).size());
    // patch the 64-bit pointer fields of GPU::Mesh / GPU::MeshLOD (SceneGPU.hpp:125-152) with this process's device addresses
    const Blob& rl = in.at("reloc");
    for (size_t i = 0; i + sizeof(Reloc) <= rl.size(); i += sizeof(Reloc)) {
      Reloc r;
      std::memcpy(&r, rl.data() + i, sizeof r);
      uint64_t v = reinterpret_cast<uint64_t>(dev.ptr.at(r.target_section)) + r.addend;
      std::memcpy(in.at(r.field_section).data() + r.field_offset, &v, 8);
    }
    for (const char* s : scene_sections)
      if (!in.at(s).empty()) HIP(hipMemcpy(dev.ptr.at(s), in.at(s).data(), in.at(s).size(), hipMemcpyHostToDevice));
    GPU::CullCamera cam;
    std::memcpy(&cam, in.at("camera").data(), sizeof cam);
    HizDesc hd, dd0, dd1;
    std::memcpy(&hd, in.at("hizdesc").data(), sizeof hd);
    std::memcpy(&dd0, in.at("depth0desc").data(), sizeof dd0);
    std::memcpy(&dd1, in.at("depth1desc").data(), sizeof dd1);
    uint32_t N;
    std::memcpy(&N, in.at("max_meshlets").data(), 4);
    const uint32_t M = cam.mesh_instance_count;
    const Blob mesh_instances0 = in.at("mesh_instances");

    RendererInstance self(0);
    dev.alloc("meshlet_instances", (uint64_t)N * 8);
    dev.alloc("visible", (uint64_t)N * 4);
    dev.alloc("reordered", (uint64_t)N * 64 * 3 * 4);
    dev.alloc("mask", in.at("mask_in").size());
    dev.alloc("hiz", hd.total_bytes);
    self.prepared_frame.mesh_instance_count = M;
    self.prepared_frame.max_meshlet_instance_count = N;
    self.prepared_frame.transforms_world_buffer = dev.buf("transforms");
    self.prepared_frame.meshes_buffer = dev.buf("meshes");
    self.prepared_frame.mesh_instances_buffer = dev.buf("mesh_instances");
    self.prepared_frame.meshlet_instances_buffer = dev.buf("meshlet_instances");
    self.prepared_frame.visible_meshlet_instances_indices_buffer = dev.buf("visible");
    self.prepared_frame.meshlet_instance_visibility_mask_buffer = dev.buf("mask");
    self.prepared_frame.reordered_indices_buffer = dev.buf("reordered");
    std::vector<std::pair<std::string, Blob>> out;
    auto dump_pass = [&](const std::string& tag, CullGeometryContext& c, bool late) {
      Blob cb = counters(self, c);
      oxc_counters k;
      std::memcpy(&k, cb.data(), sizeof k);
      const uint64_t first = late ? k.early_visible_meshlet_instances : 0;
      Blob vis = dev.download("visible", (first + k.cull_triangles_cmd_x) * 4);
      out.push_back({tag + "_counters", cb});
      out.push_back({tag + "_visible", Blob(vis.begin() + (long)(first * 4), vis.end())});
      out.push_back({tag + "_indices", dev.download("reordered", (uint64_t)k.draw_index_count * 4)});
    };

    // ---- sequence A: plain pipeline
    {
      auto cull_geometry_context = CullGeometryContext{.init_cull_meshes = true, .cull_flags = GPU::CullFlag::TestAll, .cull_camera = cam};
      self.cull_geometry(cull_geometry_context);
      dump_pass("A", cull_geometry_context, false);
      oxc_counters k;
      std::memcpy(&k, out[out.size() - 3].second.data(), sizeof k);
      out.push_back({"A_meshlet_instances", dev.download("meshlet_instances", (uint64_t)k.total_visible_meshlet_instances * 8)});
      out.push_back({"A_mesh_instances", dev.download("mesh_instances", mesh_instances0.size())});
      // the draw that consumes the lists (DrawGeometry.cpp:104-190), 512 x 384 like the raster fixture
      const uint32_t W = 512, H = 384;
      dev.alloc("visdepth", (uint64_t)W * H * 8);
      dev.alloc("draw_depth", (uint64_t)W * H * 4);
      dev.alloc("draw_vis", (uint64_t)W * H * 4);
      auto main_geometry_context = MainGeometryContext{.cull_camera = cam};
      main_geometry_context.depth_attachment = ImageAttachment{};
      main_geometry_context.depth_attachment.dptr = dev.ptr.at("draw_depth");
      main_geometry_context.depth_attachment.width = W;
      main_geometry_context.depth_attachment.height = H;
      main_geometry_context.depth_attachment.levels = 1;
      main_geometry_context.visbuffer_attachment = dev.buf("draw_vis");
      main_geometry_context.visdepth_buffer = dev.buf("visdepth");
      main_geometry_context.draw_geometry_cmd_buffer = cull_geometry_context.draw_geometry_cmd_buffer;
      self.draw_for_visbuffer(main_geometry_context);
      out.push_back({"A_visdepth", dev.download("visdepth", (uint64_t)W * H * 8)});
      out.push_back({"A_draw_depth", dev.download("draw_depth", (uint64_t)W * H * 4)});
      out.push_back({"A_draw_vis", dev.download("draw_vis", (uint64_t)W * H * 4)});
    }
    // ---- sequence B: two-pass occlusion in the reference's order; fresh instance table + the fixture's prior mask
    HIP(hipMemcpy(dev.ptr.at("mesh_instances"), mesh_instances0.data(), mesh_instances0.size(), hipMemcpyHostToDevice));
    HIP(hipMemcpy(dev.ptr.at("mask"), in.at("mask_in").data(), in.at("mask_in").size(), hipMemcpyHostToDevice));
    {
      // last frame's pyramid: built from depth0 through the same entry point
      auto main_geometry_context = MainGeometryContext{.cull_camera = cam};
      main_geometry_context.depth_attachment = image_of(dev.ptr.at("depth0"), dd0);
      main_geometry_context.hiz_attachment = image_of(dev.ptr.at("hiz"), hd);
      self.generate_hiz(main_geometry_context);
      out.push_back({"B_hiz0", dev.download("hiz", hd.total_bytes)});

      auto cull_geometry_context = CullGeometryContext{.use_hiz = true, .init_cull_meshes = true, .cull_flags = GPU::CullFlag::TestAll, .cull_camera = cam};
      const auto run_geometry_pass = [&](bool late) {  // RendererInstance.cpp:793-840
        if (late) {
          cull_geometry_context.cull_flags |= GPU::CullFlag::LatePass;
          cull_geometry_context.init_cull_meshes = false;
          cull_geometry_context.cull_camera = cam;
        }
        cull_geometry_context.hiz_attachment = main_geometry_context.hiz_attachment;
        self.cull_geometry(cull_geometry_context);
        main_geometry_context.draw_geometry_cmd_buffer = cull_geometry_context.draw_geometry_cmd_buffer;
        main_geometry_context.visibility_buffer = cull_geometry_context.visibility_buffer;
      };
      run_geometry_pass(false);
      dump_pass("B_early", cull_geometry_context, false);
      main_geometry_context.depth_attachment = image_of(dev.ptr.at("depth1"), dd1);  // "draw": the early list's depth
      self.generate_hiz(main_geometry_context);
      out.push_back({"B_hiz1", dev.download("hiz", hd.total_bytes)});
      run_geometry_pass(true);
      dump_pass("B_late", cull_geometry_context, true);
      out.push_back({"B_mask", dev.download("mask", in.at("mask_in").size())});
    }
    write_container(argv[2], out);
    std::printf("shim_frame ok: %zu outputs\n", out.size());
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "shim_frame: %s\n", e.what());
    return 1;
  }
}
