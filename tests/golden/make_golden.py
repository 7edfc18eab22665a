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
from oxylus_amd.lib import CullCamera  # noqa: E402
from oxylus_amd.renderer import HpbAttachment  # noqa: E402
from oxylus_amd.synth import SceneSpec, build_meshlets_simple, make_depth, make_mesh, make_scene, perspective_reversed_z  # noqa: E402
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


def terrain_camera():
    proj = perspective_reversed_z(60.0, 1.0, 0.1, 2000.0).view(4, 4)
    view = torch.eye(4)
    view[3, 0:3] = -torch.tensor([0.0, 40.0, 0.0])
    pv = (proj.t() @ view.t()).t().contiguous().flatten()
    cam = CullCamera()
    for i in range(16):
        cam.projection_view[i] = float(pv[i])
    cam.position[1] = 40.0
    cam.near_clip = 0.1
    return cam, pv.numpy()


def widening_rows():
    """one small fixture with inputs and expected outputs per row."""
    out = {}
    # 8f-1 bounds producer: a bumpy height field with degenerate triangles
    pos, tris = make_mesh("terrain", n=14, seed=0x601F)
    meshlets, vidx, micro = build_meshlets_simple(tris)
    b, m6, q = oracle.build_meshlet_bounds(pos, meshlets, vidx, micro)
    out.update(bounds_positions=pos.numpy(), bounds_meshlets=meshlets.numpy(), bounds_vidx=vidx.numpy(), bounds_micro=micro.numpy(),
               bounds_records=b.numpy(), bounds_mesh6=m6.numpy(), bounds_qpos=q.numpy())
    # 8f-3 HPB producer: 3 layers of 9 x 6 pages, 4 levels (odd extents)
    g = torch.Generator().manual_seed(0x6020)
    pt = torch.randint(0, 32, (3, 6, 9), generator=g, dtype=torch.int32)
    pt[torch.rand((3, 6, 9), generator=g) < 0.25] = 7
    hpb = HpbAttachment.create(9, 6, 3, 4, "cpu")
    oracle.generate_hpb(pt, oracle.make_hpb(hpb.data, 9, 6, 3, 4, hpb.level_offset))
    out.update(hpb_page_table=pt.numpy(), hpb_data=hpb.data.numpy(), hpb_level_offset=np.asarray(hpb.level_offset, dtype=np.int64))
    # 8f-4 terrain cull: 37 x 29 patches, early then late against a 64 x 64 HiZ
    g = torch.Generator().manual_seed(0x6021)
    lo = torch.rand((29, 37), generator=g) * 0.6
    mm = torch.stack([lo, lo + torch.rand((29, 37), generator=g) * 0.4], -1).contiguous()
    depth = make_depth(128, 128, 16, seed=0x6021)
    hz, levels, offs = oracle_hiz(depth, 64, 64)
    hzo = oracle.make_hiz(hz.numpy(), 64, 64, levels, offs)
    total = 37 * 29
    mask = ((torch.rand(((total + 31) // 32, 32), generator=g) < 0.35).to(torch.int64) << torch.arange(32)).sum(1).to(torch.int32)
    cam, pv = terrain_camera()
    out.update(terrain_minmax=mm.numpy(), terrain_depth=depth.numpy(), terrain_mask_in=mask.numpy().copy(), terrain_pv=pv,
               terrain_params=np.asarray([-300.0, -700.0, 600.0, 650.0, -5.0, 60.0], dtype=np.float32))
    early = oracle.cull_terrain([-300.0, -700.0], [600.0, 650.0], (37, 29), -5.0, 60.0, mm, cam, L.CULL_TEST_ALL, hzo, mask)
    late = oracle.cull_terrain([-300.0, -700.0], [600.0, 650.0], (37, 29), -5.0, 60.0, mm, cam, L.CULL_TEST_ALL | L.CULL_LATE_PASS, hzo, mask)
    out.update(terrain_early=early.numpy(), terrain_late=late.numpy(), terrain_mask_out=mask.numpy())
    np.savez_compressed(os.path.join(HERE, "widening_rows.npz"), **out)
    # 8f-2 draw consumer: the plain pipeline fixture's triangle list rasterised at 512 x 384 (expected image only:
    # the inputs are pipeline_12x40.npz)
    z = np.load(os.path.join(HERE, "pipeline_12x40.npz"))
    from util import scene_from_golden

    s, _ = scene_from_golden(os.path.join(HERE, "pipeline_12x40.npz"))
    s.mesh_instances[:, 1] = torch.from_numpy(z["plain_lod_index"])  # the LODs cull_meshes selected
    vd = torch.zeros((384, 512), dtype=torch.int64)
    oracle.draw_visbuffer(s, torch.from_numpy(z["plain_meshlet_instances"]), torch.from_numpy(z["plain_indices"]), z["camera_pv"], 512, 384, vd)
    assert int((vd != 0).sum()) >= 50
    np.savez_compressed(os.path.join(HERE, "raster_512x384.npz"), visdepth=vd.numpy())


def vertex_streams():
    """SURVEY 8(f)-1, format side: the three quantised vertex streams (specials first: signed zeros, +-1, out-of-range, Inf,
    NaN, half overflow / underflow, snorm10 ties, a float denormal) and two mesh-blob layouts."""
    g = torch.Generator().manual_seed(404)  # the data of tests/test_mesh_blob.py::test_gpu_quantize_vertex_streams_matches_checker
    V = 100_003
    pos = (torch.rand((V, 3), generator=g) * 2 - 1) * torch.tensor([1e-6, 1.0, 7e4])[torch.randint(0, 3, (V, 1), generator=g)]
    nrm = torch.nn.functional.normalize(torch.randn((V, 3), generator=g), dim=1)
    uv = torch.rand((V, 2), generator=g) * 4 - 1
    specials = torch.tensor([0.0, -0.0, 1.0, -1.0, 1.5, -1.5, float("inf"), float("-inf"), float("nan"), 65504.0, 65520.0, 6.1e-5, 5.96e-8,
                             0.5 / 511, 1.5 / 511, -0.5 / 511, 1e-40])
    for t in (pos, nrm, uv):
        t.reshape(-1)[:specials.numel()] = specials
    pos, nrm, uv = pos[:4096].contiguous(), nrm[:4096].contiguous(), uv[:4096].contiguous()
    qpos, qnrm, quv = oracle.quantize_vertex_streams(pos, nrm, uv)
    counts = [[(9, 3, 13, 7)], [(6000, 47, 5640, 2950), (3000, 24, 2880, 1500), (1500, 12, 1440, 760)]]
    layouts = []
    for (v, tex), c in zip(((5, False), (1234, True)), counts):
        lay = oracle.mesh_blob_layout(v, tex, c)
        row = [v, int(tex), len(c), lay["size"], lay["lod_metadata_offset"], lay["vertex_positions"], lay["vertex_normals"], lay["texture_coords"]]
        for cc, lo in zip(c, lay["lods"]):
            row += list(cc) + [lo[k] for k in ("indices", "meshlets", "meshlet_bounds", "local_triangle_indices", "indirect_vertex_indices")]
        layouts.append(np.asarray(row + [0] * (8 + 9 * 3 - len(row)), dtype=np.int64))
    np.savez_compressed(os.path.join(HERE, "vertex_streams_4096.npz"), positions=pos.numpy(), normals=nrm.numpy(), texcoords=uv.numpy(),
                        qpos=qpos.numpy(), qnrm=qnrm.numpy(), quv=quv.numpy(), blob_layouts=np.stack(layouts))


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
    widening_rows()
    vertex_streams()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
