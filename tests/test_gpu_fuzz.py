"""Randomised differential test of the main path: scene shape, list length, flags, pass structure and mask drawn from a seeded
generator, every output compared with the CPU checker.  Sizes straddle the work-distribution boundaries of the kernels (64-meshlet
groups, 256-meshlet wave steps, 1024-meshlet chunks, fewer / more steps than ticket counters)."""
import numpy as np
import pytest
import torch

from oxylus_amd import lib as L
from oxylus_amd.renderer import ImageAttachment
from oxylus_amd.synth import SceneSpec, make_depth, make_scene
from util import assert_same, gpu_frame, oracle_frame, oracle_hiz

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.default_rng(seed)
    k = int(rng.choice([1, 3, 7, 63, 64, 65, 100, 255, 256, 257, 1000]))
    target = int(rng.choice([1, 60, 250, 1030, 5000, 70_000, 300_000]))
    m = max(1, min(target // k, 4000))
    spec = SceneSpec(n_mesh_instances=m, meshlets_per_mesh=k, seed=1000 + seed, ragged=bool(rng.integers(2)), lod_count=int(rng.choice([1, 1, 3])),
                     scene_depth=float(rng.choice([15.0, 60.0, 200.0])), tris_per_meshlet=int(rng.choice([64, 64, 17])),
                     verts_per_meshlet=int(rng.choice([64, 64, 33])))
    mode = str(rng.choice(["plain", "plain-meshes", "hiz-two-pass", "hiz-two-pass", "hiz-no-occlusion", "hiz-late-only"]))
    mask_p = float(rng.choice([0.0, 0.3, 1.0]))
    return spec, mode, mask_p, int(rng.choice([128, 256, 512]))


@pytest.mark.parametrize("seed", range(24))
def test_random_frame_matches_checker(renderer, oracle_lib, seed):
    spec, mode, mask_p, hw = _case(seed)
    cpu = make_scene(spec, "cpu")
    gpu = cpu.to("cuda")
    n = cpu.n_meshlet_instances
    g = torch.Generator().manual_seed(seed)
    words = (n + 31) // 32
    mask = ((torch.rand((words, 32), generator=g) < mask_p).to(torch.int64) << torch.arange(32)).sum(1).to(torch.int32)
    kw = {}
    keys = ["total", "visible", "indices"]
    if mode.startswith("plain"):
        kw = dict(run_cull_meshes=mode == "plain-meshes")
        if kw["run_cull_meshes"]:
            keys = ["total", "cull_meshlets_cmd_x", "lod_index", "meshlet_instances", "visible", "indices"]
        want = oracle_frame(cpu, **kw)
        got = gpu_frame(renderer, gpu, **kw)
    else:
        depth = make_depth(2 * hw, 2 * hw, 40, seed=seed)
        data, levels, offs = oracle_hiz(depth, hw, hw)
        att = ImageAttachment.hiz(hw, hw, "cuda")
        att.data.copy_(data.cuda())
        hizd = {"data": data, "w": hw, "h": hw, "levels": levels, "offs": offs}
        flags = L.CULL_TEST_ALL
        two = True
        if mode == "hiz-no-occlusion":
            flags = L.CULL_TEST_FRUSTUM
        if mode == "hiz-late-only":
            flags, two = L.CULL_TEST_ALL | L.CULL_LATE_PASS, False
        want = oracle_frame(cpu, cull_flags=flags, use_hiz=True, hiz=hizd, mask=mask.clone(), two_pass=two)
        got = gpu_frame(renderer, gpu, cull_flags=flags, use_hiz=True, hiz=att, mask=mask.clone(), two_pass=two)
        keys = [k for k in want.keys() if k in got]
        # the same calls with share_pass_tests (include/oxcull.h): a cache between the early and the late call of a two-pass frame, ignored
        # by the other modes -- the checker's bytes either way
        shared = gpu_frame(renderer, gpu, cull_flags=flags, use_hiz=True, hiz=att, mask=mask.clone(), two_pass=two, share_pass_tests=True)
        assert_same(want, shared, keys)
    assert_same(want, got, keys)


def _tri_case(seed):
    rng = np.random.default_rng(5000 + seed)
    wide = int(rng.choice([0, 1, 2, 2]))
    tris = int(rng.choice([17, 64, 64])) if wide == 0 else int(rng.choice([64, 100, 124, 128]))
    k = int(rng.choice([5, 63, 64, 65, 130, 257]))
    m = int(rng.choice([1, 7, 40, 90]))
    share = int(rng.choice([0, 0, 2, 5]))
    spec = SceneSpec(n_mesh_instances=m, meshlets_per_mesh=k, seed=7000 + seed, ragged=bool(rng.integers(2)), scene_depth=float(rng.choice([15.0, 60.0])),
                     tris_per_meshlet=tris, verts_per_meshlet=int(rng.choice([64, 64, 33])), share_meshes=min(share, m))
    return spec, wide, int(rng.integers(2)), int(rng.choice([0, 1, 2])), bool(rng.integers(4) == 0)


@pytest.mark.parametrize("seed", range(20))
def test_random_triangle_stage_forms_match_checker(oracle_lib, seed):
    """Round 6: the index forms (packed 24 + 8, 23 + 9, {id, corner} pairs), the two list layouts (ordered / unordered_output = 1), the two load
    policies of the triangle kernels (nt / plain; 0 = by the scene) and the small-triangle cull, drawn at random over scenes of unique and of
    shared meshes; every list against the checker (unordered lists as sorted sets)."""
    import oracle
    from oxylus_amd.renderer import RendererInstance
    from util import pairs_as_u64

    spec, wide, unordered, policy, small = _tri_case(seed)
    cpu = make_scene(spec, "cpu")
    gpu = cpu.to("cuda")
    cam = cpu.cull_camera()
    want_vis = oracle.cull_meshlets(cpu, cam, cpu.meshlet_instances)
    want_idx = oracle.cull_triangles(cpu, cam, cpu.meshlet_instances, want_vis, 0, want_vis.numel(), wide=wide, small_triangle_cull=small).numpy()
    r = RendererInstance(0)
    try:
        r.debug_set_tuning(L.TUNE_TRI_LOADS, policy)
        got = gpu_frame(r, gpu, unordered_output=unordered, wide_triangle_index=wide, small_triangle_cull=small, max_tris=128 if wide else 64)
    finally:
        r.debug_set_tuning(L.TUNE_TRI_LOADS, 0)
        r.close()
    key = (lambda a: pairs_as_u64(a)) if wide == 2 else (lambda a: a.view(np.uint32))
    if unordered:
        assert np.array_equal(np.sort(got["visible"]), want_vis.numpy())
        assert np.array_equal(np.sort(key(got["indices"])), np.sort(key(want_idx)))
    else:
        assert np.array_equal(got["visible"], want_vis.numpy())
        assert np.array_equal(got["indices"], want_idx)
