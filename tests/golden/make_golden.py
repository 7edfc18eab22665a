#!/usr/bin/env python3
"""Generates the committed golden fixtures with the CPU oracle (run in the authoring container):

    python tests/golden/make_golden.py

The reference has no golden vectors for this path (parity unpinned), so these pin
the oracle against itself over time and let the GPU tests compare against data instead of code.
Each .npz holds the INPUT arrays in the reference GPU layouts (pointer-free: offsets instead of
addresses) and the EXPECTED outputs.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from oxylus_amd import lib as L  # noqa: E402
from oxylus_amd.synth import SceneSpec, make_depth, make_scene  # noqa: E402
from util import oracle_frame, oracle_hiz  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def scene_arrays(s):
    t = s._lod_tables
    return {
        "bounds": s.bounds.numpy(), "meshlets": s.meshlets.numpy(), "micro": s.micro.numpy(), "vidx": s.vidx.numpy(),
        "positions": s.positions.numpy(), "transforms": s.transforms.numpy(), "mesh_instances": s.mesh_instances.numpy(),
        "meshlet_instances": s.meshlet_instances.numpy(),
        "lods32": s.lods.view(torch.int32).numpy()[:, 10:16].copy(),     # counts + error (pointer-free part)
        "meshes32": s.meshes.view(torch.int32).numpy()[:, 6:8].copy(),  # vertex_count, lod_count
        "mesh_bounds": s.meshes.view(torch.int32).numpy()[:, 10:16].copy(),
        "meshlet_start": t["meshlet_start"].numpy(), "vidx_start": t["vidx_start"].numpy(),
        "micro_start": t["micro_start"].numpy(), "mesh_vertex_start": t["mesh_vertex_start"].numpy(),
        "camera_pv": np.asarray(s.camera["projection_view"], dtype=np.float32),
        "camera_misc": np.asarray(list(s.camera["position"]) + [s.camera["acceptable_lod_error"]] + list(s.camera["resolution"]) +
                                  [s.camera["near_clip"]], dtype=np.float32),
        "spec": np.asarray([s.spec.n_mesh_instances, s.spec.meshlets_per_mesh, s.spec.lod_count, s.n_meshes], dtype=np.int64),
    }


def main():
    oracle.build()
    # (a) full pipeline: cull_meshes + LOD, two-pass occlusion against a 64x64 HiZ, triangles
    spec = SceneSpec(n_mesh_instances=12, meshlets_per_mesh=40, lod_count=2, seed=0x601D, ragged=True, scene_depth=120.0)
    s = make_scene(spec, "cpu")
    depth = make_depth(128, 128, 16, seed=0x601D)
    hz, levels, offs = oracle_hiz(depth, 64, 64)
    arrays = scene_arrays(s)
    arrays["depth"] = depth.numpy()
    arrays["hiz"] = hz.numpy()
    arrays["hiz_offs"] = np.asarray(offs, dtype=np.int64)
    plain = oracle_frame(s.clone(), run_cull_meshes=True)
    for k in ("total", "cull_meshlets_cmd_x"):
        arrays["plain_" + k] = np.asarray(plain[k])
    for k in ("lod_index", "meshlet_instances", "visible", "indices"):
        arrays["plain_" + k] = plain[k]
    s2 = s.clone()
    g = torch.Generator().manual_seed(7)
    words = (s2.n_meshlet_instances + 31) // 32
    mask = ((torch.rand((words, 32), generator=g) < 0.3).to(torch.int64) << torch.arange(32)).sum(1).to(torch.int32)
    hizd = {"data": hz, "w": 64, "h": 64, "levels": levels, "offs": offs}
    two = oracle_frame(s2, use_hiz=True, hiz=hizd, mask=mask, two_pass=True)
    arrays["mask_in"] = mask.numpy()
    for k in ("early", "late"):
        arrays["two_" + k] = np.asarray(two[k])
    for k in ("early_visible", "late_visible", "early_indices", "late_indices", "mask"):
        arrays["two_" + k] = two[k]
    np.savez_compressed(os.path.join(HERE, "pipeline_12x40.npz"), **arrays)

    # (b) meshlet stage only, bounds-only scene, waves straddling instances
    spec = SceneSpec(n_mesh_instances=37, meshlets_per_mesh=111, seed=0x601E, with_geometry=False, nonuniform_scale=True)
    s = make_scene(spec, "cpu")
    arrays = scene_arrays(s)
    st = oracle.MarginStats()
    vis = oracle.cull_meshlets(s, s.cull_camera(), s.meshlet_instances, stats=st)
    arrays["visible"] = vis.numpy()
    arrays["near_threshold"] = np.asarray(st.meshlets_near_threshold)
    np.savez_compressed(os.path.join(HERE, "meshlets_37x111.npz"), **arrays)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
