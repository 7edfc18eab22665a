#!/usr/bin/env python3
"""tools/unpinned_gap.py [out.json] -- how large is the part of the result that NO oracle can pin?  (CPU only.)

The reference compiles its shaders with SLANG_FLOATING_POINT_MODE_FAST (ResourceCompiler/private/Session.cpp:49-58) and has no
golden vectors, so "bit-exact against the Vulkan path" can only mean: exact modulo the decisions a fast-math compiler may
legally flip.  Two measurements per scene:
  * boundary set -- elements with a comparison within 4 ulp of its threshold (orc_margin_stats);
  * fast-math envelope -- the same algorithm rebuilt with fused multiply-adds and reciprocal divisions
    (oracle/liboxcull_oracle_fast.so): how many visible meshlets / triangles differ from the canonical result.
Scenes: the committed golden fixtures and a bench-shaped synthetic scene (200 instances x 1000 meshlets, 1024^2 HiZ, p = 0.3 mask)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle  # noqa: E402
from oxylus_amd import lib as L  # noqa: E402
from oxylus_amd.synth import SceneSpec, hiz_layout, make_depth, make_scene  # noqa: E402


def run(scene, hiz_size, depth, mask0):
    cam = scene.cull_camera()
    mli = scene.meshlet_instances
    n = mli.shape[0]
    levels, offs, total = hiz_layout(hiz_size, hiz_size)
    data = torch.zeros(total // 4, dtype=torch.float32)
    oracle.generate_hiz(depth, data, hiz_size, hiz_size, levels, offs)
    hz = oracle.make_hiz(data, hiz_size, hiz_size, levels, offs)
    st_m, st_t = oracle.MarginStats(0, 0), oracle.MarginStats(0, 0)
    out = {"plain": oracle.cull_meshlets(scene, cam, mli, stats=st_m)}
    out["plain_tris"] = oracle.cull_triangles(scene, cam, mli, out["plain"], 0, out["plain"].numel(), stats=st_t)
    out["tri_boundary"] = oracle.triangle_boundary_flags(scene, cam, mli, out["plain"], 0, out["plain"].numel())
    v = oracle.Visibility(n, 0, 0)
    buf = torch.zeros(max(n, 1), dtype=torch.int32)
    mask = mask0.clone()
    st_h = oracle.MarginStats(0, 0)
    for tag, flags in (("early", L.CULL_TEST_ALL), ("late", L.CULL_TEST_ALL | L.CULL_LATE_PASS)):
        k = oracle.cull_meshlets_hiz(scene, cam, mli, flags, hz, v, mask, buf, stats=st_h)
        first = v.early if tag == "late" else 0
        out[tag] = buf[first:first + k].clone()
    out["mask"] = mask
    return out, {"meshlets_within_4ulp_plain": int(st_m.meshlets_near_threshold), "triangles_within_4ulp": int(st_t.triangles_near_threshold),
                 "meshlets_within_4ulp_two_pass": int(st_h.meshlets_near_threshold), "hiz": data}


def triangle_counts(scene, visible):
    """triangle_count of every visible meshlet (via the LOD table of its instance: the scenes here have one LOD in use)."""
    mli = scene.meshlet_instances[visible.to(torch.int64)].to(torch.int64)
    inst = scene.mesh_instances[mli[:, 0]].to(torch.int64)
    L = scene.spec.lod_count
    start = scene._lod_tables["meshlet_start"][inst[:, 0] * L + inst[:, 1]]
    return scene.meshlets[start + mli[:, 1], 3].tolist()


def sym_diff(a, b):
    return int(np.setxor1d(a.numpy(), b.numpy()).size)


def measure(name, scene, hiz_size, depth, mask0):
    canon, stats = run(scene, hiz_size, depth, mask0)
    with oracle.variant("fast"):
        fast, _ = run(scene, hiz_size, depth, mask0)
    tri = lambda t: t.view(-1, 3)[:, 0] if t.numel() else t  # noqa: E731  one packed id per triangle
    n = scene.n_meshlet_instances
    # every triangle the envelope build decides differently must sit in the conditioning-aware boundary set
    flipped = np.setxor1d(tri(canon["plain_tris"]).numpy(), tri(fast["plain_tris"]).numpy()).astype(np.int64) & 0xFFFFFFFF
    slot_of = {int(v): i for i, v in enumerate(canon["plain"].tolist())}
    fl = canon["tri_boundary"].numpy()
    outside = sum(1 for x in flipped.tolist() if not fl[slot_of[x >> 8], (x & 0xFF) // 3])
    tested = int(sum(min(int(t), 64) for t in triangle_counts(scene, canon["plain"])))
    return {"scene": name, "meshlet_instances": n, "visible_plain": int(canon["plain"].numel()), "triangles_plain": int(canon["plain_tris"].numel() // 3),
            "visible_early": int(canon["early"].numel()), "visible_late": int(canon["late"].numel()),
            "boundary_set": {**{k: v for k, v in stats.items() if k != "hiz"}, "triangles_tested": tested,
                             "triangles_ill_conditioned (|det - 1e-4| or |clip.z| <= 4 eps sum|terms|)": int(fl.sum()),
                             "envelope_flips_outside_that_set": outside},
            "fast_math_envelope": {"visible_plain_differ": sym_diff(canon["plain"], fast["plain"]), "triangles_differ": sym_diff(tri(canon["plain_tris"]), tri(fast["plain_tris"])),
                                   "visible_early_differ": sym_diff(canon["early"], fast["early"]), "visible_late_differ": sym_diff(canon["late"], fast["late"]),
                                   "mask_bits_differ": int(np.unpackbits((canon["mask"].numpy() ^ fast["mask"].numpy()).view(np.uint8)).sum())}}


def scenes():
    from util import scene_from_golden

    g = os.path.join(ROOT, "tests", "golden")
    s, z = scene_from_golden(os.path.join(g, "pipeline_12x40.npz"), "cpu")
    yield "tests/golden/pipeline_12x40.npz", s, 64, torch.from_numpy(z["depth"]), torch.from_numpy(z["mask_in"])
    s, z = scene_from_golden(os.path.join(g, "meshlets_37x111.npz"), "cpu")
    yield "tests/golden/meshlets_37x111.npz", s, 128, make_depth(256, 256, 32, seed=9), torch.zeros((s.n_meshlet_instances + 31) // 32, dtype=torch.int32)
    # a real mesh through the producer path (sphere: well-formed triangles), 300 instances
    from oxylus_amd.synth import build_meshlets_simple, make_mesh, make_scene_from_mesh

    pos, tris = make_mesh("sphere", n=24, seed=1)
    meshlets, vidx, micro = build_meshlets_simple(tris)
    bounds, mesh6, qpos = oracle.build_meshlet_bounds(pos, meshlets, vidx, micro)
    s = make_scene_from_mesh(300, bounds, meshlets, micro, vidx, qpos, mesh6, seed=77)
    yield "UV sphere (1104 triangles) x 300 instances through the bounds producer", s, 256, make_depth(512, 512, 32, seed=7), torch.zeros((s.n_meshlet_instances + 31) // 32, dtype=torch.int32)
    s = make_scene(SceneSpec(n_mesh_instances=200, meshlets_per_mesh=1000, seed=0x0A1DE5 + 2, with_geometry=True), "cpu")
    gen = torch.Generator().manual_seed(5)
    words = (s.n_meshlet_instances + 31) // 32
    bits = (torch.rand((words, 32), generator=gen) < 0.3).to(torch.int64)
    yield "synthetic 200 x 1000 (bench generator), 1024^2 HiZ, prior mask p = 0.3", s, 1024, make_depth(2048, 2048, 64, seed=3), (bits << torch.arange(32)).sum(1).to(torch.int32)


def main():
    oracle.build()
    rows = [measure(*sc) for sc in scenes()]
    doc = {"what": __doc__.split("\n\n")[1], "rows": rows}
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r02_unpinned_gap.json")
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    for r in rows:
        print(r["scene"], r["boundary_set"], r["fast_math_envelope"])


if __name__ == "__main__":
    main()
