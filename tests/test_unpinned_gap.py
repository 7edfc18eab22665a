"""The part of the result no oracle can pin: the reference is compiled with fast-math and has no golden
vectors, so parity with the Vulkan path can only hold modulo the decisions a legal re-association / fusion may flip.  These
tests tie the two measurements of tools/unpinned_gap.py together: every decision the "fast-math envelope" build of the oracle
(fused multiply-adds, reciprocal divisions) takes differently lies inside the conditioning-aware boundary set."""
import os

import numpy as np
import torch

import oracle
from oxylus_amd.synth import SceneSpec, build_meshlets_simple, make_mesh, make_scene, make_scene_from_mesh

from util import scene_from_golden

HERE = os.path.dirname(os.path.abspath(__file__))


def _check(scene):
    cam, mli = scene.cull_camera(), scene.meshlet_instances
    vis = oracle.cull_meshlets(scene, cam, mli)
    tris = oracle.cull_triangles(scene, cam, mli, vis, 0, vis.numel())
    with oracle.variant("fast"):
        vis_f = oracle.cull_meshlets(scene, cam, mli)
        tris_f = oracle.cull_triangles(scene, cam, mli, vis, 0, vis.numel())
    flags = oracle.triangle_boundary_flags(scene, cam, mli, vis, 0, vis.numel()).numpy()
    one = lambda t: (t.view(-1, 3)[:, 0].numpy().astype(np.int64) & 0xFFFFFFFF) if t.numel() else np.zeros(0, dtype=np.int64)  # noqa: E731
    flipped = np.setxor1d(one(tris), one(tris_f))
    slot = {int(v): i for i, v in enumerate(vis.tolist())}
    outside = [x for x in flipped.tolist() if not flags[slot[x >> 8], (x & 0xFF) // 3]]
    return vis, vis_f, flipped, outside, flags


def test_envelope_build_is_a_different_library_with_the_same_interface(oracle_lib):
    assert oracle.lib().orc_is_fast_envelope() == 0
    with oracle.variant("fast"):
        assert oracle.lib().orc_is_fast_envelope() == 1
        # an exactly representable case gives the same answer in both builds
        assert oracle.test_frustum(np.eye(4, dtype=np.float32).reshape(-1), [0, 0, 0.5], [0.5, 0.5, 0.5])
    assert oracle.lib().orc_is_fast_envelope() == 0


def test_fast_math_flips_lie_inside_the_boundary_set_golden_scene(oracle_lib):
    s, _ = scene_from_golden(os.path.join(HERE, "golden", "pipeline_12x40.npz"), "cpu")
    vis, vis_f, flipped, outside, flags = _check(s)
    assert torch.equal(vis, vis_f)            # no meshlet decision moves in this fixture
    assert flipped.size > 0                   # ... but triangle decisions do: the backface determinant cancels
    assert not outside, f"{len(outside)} flipped triangles are not flagged as boundary cases"
    assert flags.sum() < flags.size           # the set is not "everything"


def test_fast_math_flips_lie_inside_the_boundary_set_real_mesh(oracle_lib):
    pos, tris = make_mesh("sphere", n=16, seed=3)
    meshlets, vidx, micro = build_meshlets_simple(tris)
    bounds, mesh6, qpos = oracle.build_meshlet_bounds(pos, meshlets, vidx, micro)
    s = make_scene_from_mesh(120, bounds, meshlets, micro, vidx, qpos, mesh6, seed=79)
    vis, vis_f, flipped, outside, flags = _check(s)
    assert not outside
    # well-formed triangles: the boundary set is a small minority, unlike the random triangle soup of the bench generator
    assert 0 < flags.sum() < 0.2 * (vis.numel() * 64)


def test_committed_report_is_current_in_shape():
    import json

    doc = json.load(open(os.path.join(os.path.dirname(HERE), "profiles", "r02_unpinned_gap.json")))
    assert len(doc["rows"]) >= 3
    for row in doc["rows"]:
        assert row["boundary_set"]["envelope_flips_outside_that_set"] == 0
        assert row["fast_math_envelope"]["visible_plain_differ"] == 0
