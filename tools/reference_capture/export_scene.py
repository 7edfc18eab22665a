#!/usr/bin/env python3
"""tools/reference_capture/export_scene.py <golden.npz> <out_dir> -- writes a golden scene (tests/golden/*.npz, the layout of
tests/golden/make_golden.py) as raw little-endian buffers in the reference's GPU layouts plus manifest.json, for a harness on the
Vulkan side to upload.  The .npz is pointer-free; device addresses are patched by the uploader as upload_gltf_mesh does
(AssetManager_GLTF.cpp:780-800): manifest.json lists, per GPU::MeshLOD / GPU::Mesh record, which buffer and element each pointer
field refers to.

Buffers (Oxylus/include/Scene/SceneGPU.hpp:84-152, scalar layout):
  meshlet_bounds.bin   GPU::MeshletBounds[]   16 B      meshlets.bin   GPU::Meshlet[] 16 B
  micro.bin            u8 local_triangle_indices         vidx.bin       u32 indirect_vertex_indices
  positions.bin        u16x4 vertex_positions            transforms.bin GPU::TransformWorld[] 64 B (column-major mat4)
  mesh_instances.bin   GPU::MeshInstance[]   20 B        meshlet_instances.bin GPU::MeshletInstance[] 8 B (the list cull_meshes would build)
  depth.bin            R32F depth image the pyramid is built from (reversed Z, 0 = far)
  mask_in.bin          u32 meshlet_instance_visibility_mask before the early pass
  camera.json          GPU::CullCamera fields"""
import json
import os
import sys

import numpy as np


def main():
    src, out = sys.argv[1], sys.argv[2]
    os.makedirs(out, exist_ok=True)
    z = np.load(src)
    names = {"bounds": "meshlet_bounds.bin", "meshlets": "meshlets.bin", "micro": "micro.bin", "vidx": "vidx.bin", "positions": "positions.bin",
             "transforms": "transforms.bin", "mesh_instances": "mesh_instances.bin", "meshlet_instances": "meshlet_instances.bin"}
    manifest = {"source": os.path.basename(src), "buffers": {}}
    for key, fname in names.items():
        a = np.ascontiguousarray(z[key])
        a.tofile(os.path.join(out, fname))
        manifest["buffers"][fname] = {"bytes": int(a.nbytes), "dtype": str(a.dtype), "shape": list(a.shape)}
    for key in ("depth", "mask_in"):
        if key in z.files:
            a = np.ascontiguousarray(z[key])
            a.tofile(os.path.join(out, key + ".bin"))
            manifest["buffers"][key + ".bin"] = {"bytes": int(a.nbytes), "dtype": str(a.dtype), "shape": list(a.shape)}
    M, K, lod_count, n_meshes = [int(x) for x in z["spec"]]
    lods32, meshes32 = z["lods32"], z["meshes32"]
    # GPU::Mesh {u64 vertex_positions, vertex_normals, texture_coords; u32 vertex_count, lod_count; u64 lods; f32x3 aabb_center, aabb_extent}
    # GPU::MeshLOD {u64 indices, meshlets, meshlet_bounds, local_triangle_indices, indirect_vertex_indices; u32 counts[5]; f32 error}
    meshes = []
    for m in range(n_meshes):
        lods = []
        for l in range(int(meshes32[m, 1])):
            row = m * lod_count + l
            lods.append({"meshlets": {"buffer": "meshlets.bin", "element": int(z["meshlet_start"][row]), "stride": 16},
                         "meshlet_bounds": {"buffer": "meshlet_bounds.bin", "element": int(z["meshlet_start"][row]), "stride": 16},
                         "local_triangle_indices": {"buffer": "micro.bin", "element": int(z["micro_start"][row]), "stride": 1},
                         "indirect_vertex_indices": {"buffer": "vidx.bin", "element": int(z["vidx_start"][row]), "stride": 4},
                         "counts_and_error_u32x6": [int(x) for x in lods32[row]]})
        meshes.append({"vertex_positions": {"buffer": "positions.bin", "element": int(z["mesh_vertex_start"][m]), "stride": 8},
                       "vertex_count": int(meshes32[m, 0]), "lod_count": int(meshes32[m, 1]), "aabb_center_extent_f32_bits": [int(x) for x in z["mesh_bounds"][m]],
                       "lods": lods})
    manifest["meshes"] = meshes
    misc = z["camera_misc"]
    cam = {"projection_view_column_major": [float(x) for x in z["camera_pv"]], "position": [float(x) for x in misc[0:3]], "acceptable_lod_error": float(misc[3]),
           "resolution": [float(misc[4]), float(misc[5])], "near_clip": float(misc[6]), "mesh_instance_count": M}
    json.dump(cam, open(os.path.join(out, "camera.json"), "w"), indent=1)
    manifest["hiz"] = {"extent": [64, 64], "levels": 7, "built_from": "depth.bin (128 x 128) by the engine's own hiz pass"}
    manifest["mask_words"] = int((z["meshlet_instances"].shape[0] + 31) // 32)
    json.dump(manifest, open(os.path.join(out, "manifest.json"), "w"), indent=1)
    print(f"wrote {len(manifest['buffers'])} buffers + manifest.json + camera.json to {out}")


if __name__ == "__main__":
    main()
