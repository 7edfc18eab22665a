"""GPU: unordered_output (include/oxcull.h) -- the reference's own slot allocation (atomic_add on the counter: cull_meshlets.slang:55-70,
cull_triangles.slang:71-88), aggregated per block through the ballots, one launch per stage; the HiZ meshlet stage keeps its ordered emit and
its ascending list.

SURVEY 8c(1): "counts equal and sorted index arrays byte-identical".  The ordered form emits ascending lists (packed triangle indices
ascend with (meshlet instance, triangle, corner)), so every list of an unordered call, sorted, must be the checker's bytes; the mask does
not depend on the order at all; a triangle's three packed indices stay next to each other."""
import numpy as np
import pytest
import torch

from oxylus_amd import lib as L
from oxylus_amd.renderer import CullGeometryContext, ImageAttachment, MainGeometryContext, PreparedFrame
from oxylus_amd.synth import SceneSpec, make_depth, make_scene

from util import assert_same, assert_triangles_adjacent, gpu_frame, oracle_frame, oracle_hiz, pairs_as_u64, sorted_lists

pytestmark = pytest.mark.gpu

HIZ_KEYS = ["total", "early", "late", "early_emitted", "late_emitted", "early_visible", "late_visible", "early_indices", "late_indices", "mask"]


def _hiz_setup(renderer, spec, hw, p_mask, seed):
    cpu = make_scene(spec, "cpu")
    gpu = cpu.to("cuda")
    depth = make_depth(2 * hw, 2 * hw, 48, seed=seed, device="cuda")
    hiz = ImageAttachment.hiz(hw, hw, "cuda")
    renderer.generate_hiz(MainGeometryContext(ImageAttachment.depth(depth), hiz))
    want_hiz, levels, offs = oracle_hiz(depth.cpu(), hw, hw)
    n = cpu.n_meshlet_instances
    g = torch.Generator().manual_seed(seed)
    words = max((n + 31) // 32, 1)
    bits = (torch.rand((words, 32), generator=g) < p_mask).to(torch.int64)
    mask = (bits << torch.arange(32)).sum(1).to(torch.int32)
    return cpu, gpu, hiz, {"data": want_hiz, "w": hw, "h": hw, "levels": levels, "offs": offs}, mask


@pytest.mark.parametrize("m,k,seed", [
    (1000, 1000, 3),   # configs[1]'s shape: 1 M meshlets, one block iteration per block
    (300, 37, 4),      # several mesh instances per wave step
    (5, 333, 5),       # ragged: 1665 meshlets, the last block is partial
    (1, 40, 6),        # less than one wave
], ids=["config2-shape", "many-instances-per-step", "ragged", "tiny"])
def test_plain_pipeline_unordered_is_the_ordered_set(renderer, oracle_lib, m, k, seed):
    """cull_meshlets + cull_triangles with unordered_output = 1: TWO launches after prepare (test + append, test + expand) instead of four."""
    spec = SceneSpec(n_mesh_instances=m, meshlets_per_mesh=k, with_geometry=True, seed=seed)
    cpu = make_scene(spec, "cpu")
    gpu = cpu.to("cuda")
    want = oracle_frame(cpu)
    renderer.profile_begin()
    got = gpu_frame(renderer, gpu, unordered_output=1)
    ran = renderer.profile_end()["kernels"]
    # the unordered path really ran (a silent fall-back to the ordered kernels would pass every comparison below): no emit launch at all
    assert "cull_meshlets_emit" not in ran and "cull_triangles_emit" not in ran, sorted(ran)
    assert ran["cull_meshlets_test"]["launches"] == 1 and ran["cull_triangles_test"]["launches"] == 1, ran
    assert got["visible"].size == want["visible"].size and got["indices"].size == want["indices"].size
    assert_same(want, sorted_lists(got), ["visible", "indices"])
    assert_triangles_adjacent(got["indices"])
    renderer.profile_begin()
    ordered = gpu_frame(renderer, gpu)
    ran0 = renderer.profile_end()["kernels"]
    assert ran0["cull_meshlets_emit"]["launches"] == 1 and ran0["cull_triangles_emit"]["launches"] == 1, ran0
    assert_same(want, ordered, ["visible", "indices"])  # the default stays ascending, unsorted comparison
    if m >= 300:  # the unordered list really is in another order (runs of different blocks land in arrival order) ... usually
        assert want["visible"].size > 64


@pytest.mark.parametrize("m,k,hw,p_mask,seed,share", [
    (300, 1000, 1024, 0.3, 11, False),  # the bench's shape
    (300, 1000, 1024, 0.3, 11, True),   # ... with the late call reusing the early call's camera tests
    (1500, 37, 512, 0.3, 12, False),    # many instances per wave step
    (7, 333, 256, 0.5, 13, True),       # ragged
    (3, 70, 256, 1.0, 14, False),       # less than one step, everything visible last frame
    (40, 1000, 1024, 0.0, 15, False),   # nothing visible last frame: the early call emits nothing
], ids=["bench-shape", "bench-shape-shared", "many-instances-per-step", "ragged-shared", "tiny", "cold-mask"])
def test_two_pass_hiz_frame_unordered_is_the_ordered_set(renderer, oracle_lib, m, k, hw, p_mask, seed, share):
    spec = SceneSpec(n_mesh_instances=m, meshlets_per_mesh=k, with_geometry=True, seed=seed)
    cpu, gpu, hiz, ohiz, mask = _hiz_setup(renderer, spec, hw, p_mask, seed)
    want = oracle_frame(cpu, use_hiz=True, hiz=ohiz, mask=mask, two_pass=True)
    renderer.profile_begin()
    got = gpu_frame(renderer, gpu, use_hiz=True, hiz=hiz, mask=mask, two_pass=True, unordered_output=1, share_pass_tests=share)
    ran = renderer.profile_end()["kernels"]
    # the fused triangle kernel ran (no triangle emit launch); the HiZ meshlet stage keeps its ordered emit
    assert "cull_triangles_emit" not in ran and "cull_triangles_emit_late" not in ran, sorted(ran)
    assert "cull_meshlets_emit" in ran and "cull_meshlets_emit_late" in ran, sorted(ran)
    assert_same(want, sorted_lists(got), HIZ_KEYS)
    for tag in ("early", "late"):
        if got[f"{tag}_indices"].size:
            assert_triangles_adjacent(got[f"{tag}_indices"])
    assert_same(want, got, ["early_visible", "late_visible"])  # the visible lists are ascending as they stand
    assert got["share_modes"] == ([1, 3] if share else [0, 0])
    # a second frame on the same seeded context re-zeroes what the appending kernels add to
    again = gpu_frame(renderer, gpu, use_hiz=True, hiz=hiz, mask=mask, two_pass=True, unordered_output=1, share_pass_tests=share)
    assert_same(want, sorted_lists(again), HIZ_KEYS)


def test_fused_triangle_kernel_with_more_spans_than_blocks_draws_its_last_chunks(renderer, oracle_lib):
    """Round 5: the chunks of the last, partial round of the fused kernel's grid are handed out by ticket (tris_fused_body).  With the
    grid capped at ONE block per CU (OXC_TUNE_TRI_BLOCKS_PER_CU) a small scene has several rounds of spans and a drawn remainder; the
    list, sorted, must still be the ordered form's."""
    spec = SceneSpec(n_mesh_instances=400, meshlets_per_mesh=1000, with_geometry=True, seed=41)
    cpu = make_scene(spec, "cpu")
    gpu = cpu.to("cuda")
    want = oracle_frame(cpu)
    renderer.debug_set_tuning(L.TUNE_TRI_BLOCKS_PER_CU, 1)
    try:
        got = gpu_frame(renderer, gpu, unordered_output=1)
    finally:
        renderer.debug_set_tuning(L.TUNE_TRI_BLOCKS_PER_CU, 8)
    assert want["visible"].size > 3 * 256 * 128  # more spans than three rounds of one block per CU
    assert_same(want, sorted_lists(got), ["visible", "indices"])
    assert_triangles_adjacent(got["indices"])


def test_unordered_output_values_other_than_0_and_1_are_refused(renderer):
    """Round 4's mode 2 (the HiZ meshlet tests appending per wave step) was a measured loss and left the boundary in round 5."""
    spec = SceneSpec(n_mesh_instances=4, meshlets_per_mesh=64, with_geometry=True, seed=3)
    gpu = make_scene(spec, "cuda")
    frame = PreparedFrame.create(gpu, with_triangles=True)
    renderer.prepared_frame = frame
    ctx = CullGeometryContext(use_hiz=False, init_cull_meshes=False, cull_flags=L.CULL_TEST_ALL, cull_camera=gpu.cull_camera(), stages=L.STAGE_ALL, unordered_output=2)
    renderer.seed_meshlet_instances(ctx, gpu.n_meshlet_instances)
    with pytest.raises(L.OxcError, match="unordered_output"):
        renderer.cull_geometry(ctx)


def test_repeated_calls_on_one_seeded_context_restart_their_counters(renderer, oracle_lib):
    """The appending kernels ADD to visibility.early / .late, cull_triangles_cmd.x and index_count: every call must start them from
    zero, also on a sequence that is culled again and again (the bench's loop) and when ordered and unordered calls alternate."""
    spec = SceneSpec(n_mesh_instances=120, meshlets_per_mesh=500, with_geometry=True, seed=23)
    cpu, gpu, hiz, ohiz, mask = _hiz_setup(renderer, spec, 512, 0.3, 23)
    want = oracle_frame(cpu, use_hiz=True, hiz=ohiz, mask=mask, two_pass=True)
    frame = PreparedFrame.create(gpu, with_triangles=True)
    renderer.prepared_frame = frame
    ctx = CullGeometryContext(use_hiz=True, init_cull_meshes=False, cull_flags=L.CULL_TEST_ALL, cull_camera=gpu.cull_camera(), hiz_attachment=hiz, stages=L.STAGE_ALL)
    renderer.seed_meshlet_instances(ctx, gpu.n_meshlet_instances)
    mask_gpu = mask.cuda()
    for rep, mode in enumerate([1, 1, 0, 1, 0]):
        frame.meshlet_instance_visibility_mask_buffer.copy_(mask_gpu)
        ctx.unordered_output = mode
        for tag, flags in (("early", L.CULL_TEST_ALL), ("late", L.CULL_TEST_ALL | L.CULL_LATE_PASS)):
            ctx.cull_flags = flags
            renderer.cull_geometry(ctx)
            c = renderer.read_counters(ctx)
            first = c.early_visible_meshlet_instances if tag == "late" else 0
            assert c.cull_triangles_cmd_x == want[f"{tag}_emitted"], (rep, mode, tag)
            vis = np.sort(frame.visible_meshlet_instances_indices_buffer[first:first + c.cull_triangles_cmd_x].cpu().numpy())
            assert np.array_equal(vis, want[f"{tag}_visible"]), (rep, mode, tag)
            idx = np.sort(frame.reordered_indices_buffer[:c.draw_index_count].cpu().numpy().view(np.uint32))
            assert np.array_equal(idx, want[f"{tag}_indices"].view(np.uint32)), (rep, mode, tag)
        assert (c.early_visible_meshlet_instances, c.late_visible_meshlet_instances) == (want["early"], want["late"])
        assert np.array_equal(frame.meshlet_instance_visibility_mask_buffer.cpu().numpy(), want["mask"])


def test_with_cull_meshes_and_lod_select(renderer, oracle_lib):
    """The whole cull_geometry call of the reference -- cull_meshes (frustum + LOD select) -> cull_meshlets -> cull_triangles -- unordered."""
    spec = SceneSpec(n_mesh_instances=400, meshlets_per_mesh=300, lod_count=3, with_geometry=True, seed=21)
    cpu = make_scene(spec, "cpu")
    gpu = cpu.to("cuda")
    want = oracle_frame(cpu, run_cull_meshes=True)
    got = gpu_frame(renderer, gpu, run_cull_meshes=True, unordered_output=1)
    assert_same(want, sorted_lists(got), ["total", "lod_index", "meshlet_instances", "cull_meshlets_cmd_x", "visible", "indices"])
    assert 0 < want["total"] < spec.n_mesh_instances * spec.meshlets_per_mesh


@pytest.mark.parametrize("wide,small", [(True, False), (False, True), (True, True)], ids=["wide", "small-triangle", "wide+small"])
def test_extension_variants_of_the_fused_triangle_kernel(renderer, oracle_lib, wide, small):
    import oracle

    spec = SceneSpec(n_mesh_instances=60, meshlets_per_mesh=200, with_geometry=True, seed=31, tris_per_meshlet=124 if wide else 64)
    cpu = make_scene(spec, "cpu")
    gpu = cpu.to("cuda")
    cam = cpu.cull_camera()
    want_vis = oracle.cull_meshlets(cpu, cam, cpu.meshlet_instances)
    want_idx = oracle.cull_triangles(cpu, cam, cpu.meshlet_instances, want_vis, 0, want_vis.numel(), wide=wide, small_triangle_cull=small).numpy()
    got = gpu_frame(renderer, gpu, unordered_output=1, wide_triangle_index=wide, small_triangle_cull=small, max_tris=128 if wide else 64)
    assert np.array_equal(np.sort(got["visible"]), want_vis.numpy())
    assert np.array_equal(np.sort(got["indices"].view(np.uint32)), want_idx.view(np.uint32))
    assert_triangles_adjacent(got["indices"], 9 if wide else 8)
    assert want_idx.size > 3000


def test_empty_lists(renderer, oracle_lib):
    """Nothing visible: camera looking away.  Counters are zero, nothing is written, and a visible set that is not a multiple of a span."""
    spec = SceneSpec(n_mesh_instances=20, meshlets_per_mesh=100, with_geometry=True, seed=41)
    gpu = make_scene(spec, "cpu").to("cuda")
    frame = PreparedFrame.create(gpu, with_triangles=True)
    frame.reordered_indices_buffer.fill_(-1)
    frame.visible_meshlet_instances_indices_buffer.fill_(-1)
    renderer.prepared_frame = frame
    cam = gpu.cull_camera()
    for col in range(4):  # clip.x += 1e7 * clip.w: in front of the camera everything is beyond the right plane, behind it beyond the left one
        cam.projection_view[col * 4 + 0] += 1e7 * cam.projection_view[col * 4 + 3]
    ctx = CullGeometryContext(init_cull_meshes=False, cull_flags=L.CULL_TEST_ALL, cull_camera=cam, stages=L.STAGE_ALL, unordered_output=1)
    renderer.seed_meshlet_instances(ctx, gpu.n_meshlet_instances)
    renderer.cull_geometry(ctx)
    c = renderer.read_counters(ctx)
    ctx0 = CullGeometryContext(init_cull_meshes=False, cull_flags=L.CULL_TEST_ALL, cull_camera=cam, stages=L.STAGE_ALL)
    renderer.seed_meshlet_instances(ctx0, gpu.n_meshlet_instances)
    renderer.cull_geometry(ctx0)
    c0 = renderer.read_counters(ctx0)
    assert (c.cull_triangles_cmd_x, c.draw_index_count) == (c0.cull_triangles_cmd_x, c0.draw_index_count) == (0, 0)
    assert int((frame.reordered_indices_buffer != -1).sum()) == 0 and int((frame.visible_meshlet_instances_indices_buffer != -1).sum()) == 0


def test_bad_value_is_refused(renderer):
    spec = SceneSpec(n_mesh_instances=2, meshlets_per_mesh=10, with_geometry=True, seed=1)
    gpu = make_scene(spec, "cpu").to("cuda")
    renderer.prepared_frame = PreparedFrame.create(gpu, with_triangles=True)
    ctx = CullGeometryContext(init_cull_meshes=False, cull_flags=L.CULL_TEST_ALL, cull_camera=gpu.cull_camera(), stages=L.STAGE_ALL)
    renderer.seed_meshlet_instances(ctx, gpu.n_meshlet_instances)
    ctx.unordered_output = 3
    with pytest.raises(L.OxcError) as ei:
        renderer.cull_geometry(ctx)
    assert ei.value.status == L.OXC_INVALID_ARG


@pytest.mark.parametrize("wide", [0, 1, 2], ids=["packed", "wide9", "pairs"])
def test_triangle_load_policy_changes_no_byte(oracle_lib, wide):
    """Round 6: geometry shared between instances is read with plain loads, unique geometry with `nt` loads (OXC_TUNE_TRI_LOADS: 0 = by the scene, 1 = nt,
    2 = plain).  A cache policy: every list of the ordered and of the fused form must come out the same under both, on a scene of unique meshes and on
    an instanced one (share_meshes: 3 meshes, 40 instances -- which the default picks plain loads for)."""
    import oracle
    from oxylus_amd.renderer import RendererInstance

    r = RendererInstance(0)
    try:
        for share in (0, 3):
            spec = SceneSpec(n_mesh_instances=40, meshlets_per_mesh=120, with_geometry=True, seed=71 + share, tris_per_meshlet=124 if wide else 64, share_meshes=share)
            cpu = make_scene(spec, "cpu")
            gpu = cpu.to("cuda")
            cam = cpu.cull_camera()
            want_vis = oracle.cull_meshlets(cpu, cam, cpu.meshlet_instances)
            want_idx = oracle.cull_triangles(cpu, cam, cpu.meshlet_instances, want_vis, 0, want_vis.numel(), wide=wide).numpy()
            assert want_idx.size > 2000
            key = (lambda a: pairs_as_u64(a)) if wide == 2 else (lambda a: a.view(np.uint32))
            for policy in (0, 1, 2):
                r.debug_set_tuning(L.TUNE_TRI_LOADS, policy)
                got = gpu_frame(r, gpu, unordered_output=0, wide_triangle_index=wide, max_tris=128 if wide else 64)
                assert np.array_equal(got["visible"], want_vis.numpy()) and np.array_equal(got["indices"], want_idx), (share, policy)
                # the host's own choice: 40 instances of 3 meshes are shared geometry (plain loads), 40 instances of 40 meshes are not
                assert r.debug_tri_loads_mode() == ((2 if share else 1) if policy == 0 else policy)
                gotu = gpu_frame(r, gpu, unordered_output=1, wide_triangle_index=wide, max_tris=128 if wide else 64)
                assert np.array_equal(np.sort(key(gotu["indices"])), np.sort(key(want_idx))), (share, policy)
    finally:
        r.debug_set_tuning(L.TUNE_TRI_LOADS, 0)
        r.close()
