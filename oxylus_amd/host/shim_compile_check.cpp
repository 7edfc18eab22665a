// Compile-and-link check of the C++ shim: the call sequence of RendererInstance::render's 3D
// pass (Oxylus/src/Render/RendererInstance.cpp:783-884) written against ox::amd.  Built by
// tests/test_abi.py with g++ against liboxcull.so; it is only RUN on a GPU box.
#include <cstdio>

#include "RendererInstance.hpp"

using namespace ox::amd;

int main(int argc, char**) {
  if (argc < 2) {
    std::puts("shim links (pass any argument on a GPU box to create a context)");
    return 0;
  }
  RendererInstance self(0);
  auto cull_camera = GPU::CullCamera{};
  cull_camera.mesh_instance_count = self.prepared_frame.mesh_instance_count;
  auto cull_geometry_context = CullGeometryContext{.use_hiz = true, .init_cull_meshes = true, .cull_camera = cull_camera};
  auto main_geometry_context = MainGeometryContext{.cull_camera = cull_camera};
  const auto run_geometry_pass = [&](bool late) {
    if (late) {
      cull_geometry_context.cull_flags |= GPU::CullFlag::LatePass;
      cull_geometry_context.init_cull_meshes = false;
      cull_geometry_context.cull_camera = cull_camera;
    }
    cull_geometry_context.hiz_attachment = main_geometry_context.hiz_attachment;
    try {
      self.cull_geometry(cull_geometry_context);
    } catch (const std::exception& e) {
      std::printf("expected (no buffers bound): %s\n", e.what());
    }
    main_geometry_context.draw_geometry_cmd_buffer = cull_geometry_context.draw_geometry_cmd_buffer;
    main_geometry_context.visibility_buffer = cull_geometry_context.visibility_buffer;
  };
  run_geometry_pass(false);
  try {
    self.generate_hiz(main_geometry_context);
  } catch (const std::exception& e) {
    std::printf("expected (no attachments bound): %s\n", e.what());
  }
  run_geometry_pass(true);
  return 0;
}
