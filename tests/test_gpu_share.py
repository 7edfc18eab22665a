"""GPU: share_pass_tests (include/oxcull.h) -- the late HiZ call of a frame takes the frustum + cone results from the early call of the
same frame instead of testing again.  A cache: no output byte may change, whatever the scene, and the library has to fall back to
testing whenever the late call is not the continuation of that early call."""
import numpy as np
import pytest
import torch

from oxylus_amd import lib as L
from oxylus_amd.renderer import ImageAttachment, MainGeometryContext
from oxylus_amd.synth import SceneSpec, make_depth, make_scene

from util import assert_same, gpu_frame, oracle_frame, oracle_hiz

pytestmark = pytest.mark.gpu

KEYS = ["total", "early", "late", "early_emitted", "late_emitted", "early_visible", "late_visible", "early_indices", "late_indices", "mask"]


def _setup(renderer, spec, hw, p_mask, seed):
    cpu = make_scene(spec, "cpu")
    gpu = cpu.to("cuda")
    depth = make_depth(2 * hw, 2 * hw, 48, seed=seed, device="cuda")
    hiz = ImageAttachment.hiz(hw, hw, "cuda")
    renderer.generate_hiz(MainGeometryContext(ImageAttachment.depth(depth), hiz))
    want_hiz, levels, offs = oracle_hiz(depth.cpu(), hw, hw)
    n = cpu.n_meshlet_instances
    g = torch.Generator().manual_seed(seed)
    words = (n + 31) // 32
    bits = (torch.rand((max(words, 1), 32), generator=g) < p_mask).to(torch.int64)
    mask = (bits << torch.arange(32)).sum(1).to(torch.int32)[:max(words, 1)]
    return cpu, gpu, hiz, {"data": want_hiz, "w": hw, "h": hw, "levels": levels, "offs": offs}, mask


@pytest.mark.parametrize("m,k,hw,p_mask,seed", [
    (300, 1000, 1024, 0.3, 11),   # the bench's shape: four wave steps per instance
    (1500, 37, 512, 0.3, 12),     # many instances per wave step: several rounds, runs of mask bits that are not one run
    (7, 333, 256, 0.5, 13),       # ragged: N = 2331, the last step is partial
    (3, 70, 256, 1.0, 14),        # less than one step, everything visible last frame
    (40, 1000, 1024, 0.0, 15),    # nothing visible last frame: the early call only fills the bits
    (900, 256, 512, 0.3, 16),     # instances are exactly one step
], ids=["bench-shape", "many-instances-per-step", "ragged", "tiny", "cold-mask", "one-step-instances"])
def test_late_call_reusing_the_early_tests_writes_the_checkers_bytes(renderer, oracle_lib, m, k, hw, p_mask, seed):
    spec = SceneSpec(n_mesh_instances=m, meshlets_per_mesh=k, with_geometry=True, seed=seed)
    cpu, gpu, hiz, ohiz, mask = _setup(renderer, spec, hw, p_mask, seed)
    want = oracle_frame(cpu, use_hiz=True, hiz=ohiz, mask=mask, two_pass=True)
    got = gpu_frame(renderer, gpu, use_hiz=True, hiz=hiz, mask=mask, two_pass=True, share_pass_tests=True)
    assert_same(want, got, KEYS)
    assert got["share_modes"] == [1, 3]  # the early call published (and prepared for both), the late call reused
    plain = gpu_frame(renderer, gpu, use_hiz=True, hiz=hiz, mask=mask, two_pass=True)
    assert_same(plain, got, KEYS)
    assert plain["share_modes"] == [0, 0]
    assert want["early"] + want["late"] > 0 or p_mask == 0.0


def test_with_cull_meshes_in_the_early_call(renderer, oracle_lib):
    """The early call builds the MeshletInstance list (frustum + LOD select per mesh instance), the late call continues the sequence."""
    spec = SceneSpec(n_mesh_instances=400, meshlets_per_mesh=300, lod_count=3, with_geometry=True, seed=21)
    cpu, gpu, hiz, ohiz, mask = _setup(renderer, spec, 512, 0.3, 21)
    want = oracle_frame(cpu, use_hiz=True, hiz=ohiz, mask=mask, two_pass=True, run_cull_meshes=True)
    got = gpu_frame(renderer, gpu, use_hiz=True, hiz=hiz, mask=mask, two_pass=True, run_cull_meshes=True, share_pass_tests=True)
    assert_same(want, got, KEYS + ["lod_index", "meshlet_instances", "cull_meshlets_cmd_x"])
    assert 0 < want["total"] and got["share_modes"] == [1, 3]


def _moved(cam):
    for c in range(4):
        cam.projection_view[c * 4 + 0] *= 0.8
    cam.position[0] += 3.0
    return cam


def test_falls_back_when_the_late_call_is_not_the_continuation(renderer, oracle_lib):
    """A late call with another camera, a late call without an early call, and a late call after the list was re-seeded all have the
    flag set -- and must test for themselves (the result of the same calls without the flag)."""
    spec = SceneSpec(n_mesh_instances=200, meshlets_per_mesh=500, with_geometry=True, seed=31)
    cpu, gpu, hiz, ohiz, mask = _setup(renderer, spec, 512, 0.3, 31)

    def other_camera_for_late(i, ctx):
        if i == 1:
            ctx.cull_camera = _moved(gpu.cull_camera())

    a = gpu_frame(renderer, gpu, use_hiz=True, hiz=hiz, mask=mask, two_pass=True, share_pass_tests=True, before_pass=other_camera_for_late)
    b = gpu_frame(renderer, gpu, use_hiz=True, hiz=hiz, mask=mask, two_pass=True, before_pass=other_camera_for_late)
    assert_same(b, a, KEYS)
    assert a["share_modes"] == [1, 0]
    same_cam = gpu_frame(renderer, gpu, use_hiz=True, hiz=hiz, mask=mask, two_pass=True)
    assert not np.array_equal(same_cam["late_visible"], a["late_visible"])  # the other camera does see other meshlets
    # late call only
    a = gpu_frame(renderer, gpu, cull_flags=L.CULL_TEST_ALL | L.CULL_LATE_PASS, use_hiz=True, hiz=hiz, mask=mask, share_pass_tests=True)
    b = gpu_frame(renderer, gpu, cull_flags=L.CULL_TEST_ALL | L.CULL_LATE_PASS, use_hiz=True, hiz=hiz, mask=mask)
    assert_same(b, a, ["total", "late", "late_emitted", "late_visible", "late_indices", "mask"])
    assert a["share_modes"] == [0]

    # the list is re-seeded with another length between the two calls (same buffers, same camera)
    def reseed(i, ctx):
        if i == 1:
            renderer.seed_meshlet_instances(ctx, gpu.n_meshlet_instances - 700)

    a = gpu_frame(renderer, gpu, use_hiz=True, hiz=hiz, mask=mask, two_pass=True, share_pass_tests=True, before_pass=reseed)
    b = gpu_frame(renderer, gpu, use_hiz=True, hiz=hiz, mask=mask, two_pass=True, before_pass=reseed)
    assert_same(b, a, ["late_emitted", "late_visible", "late_indices", "mask"])
    assert a["share_modes"] == [1, 0]


def test_two_frames_back_to_back_keep_their_own_bits(renderer, oracle_lib):
    """Frame A early, frame A late, frame B (other camera) early, frame B late on one context: each late call reuses its own frame's bits."""
    spec = SceneSpec(n_mesh_instances=250, meshlets_per_mesh=400, with_geometry=True, seed=41)
    cpu, gpu, hiz, ohiz, mask = _setup(renderer, spec, 512, 0.3, 41)
    cams = {}

    def camera_b(i, ctx):
        ctx.cull_camera = cams.setdefault("b", _moved(gpu.cull_camera()))

    a1 = gpu_frame(renderer, gpu, use_hiz=True, hiz=hiz, mask=mask, two_pass=True, share_pass_tests=True)
    b1 = gpu_frame(renderer, gpu, use_hiz=True, hiz=hiz, mask=mask, two_pass=True, share_pass_tests=True, before_pass=camera_b)
    a0 = gpu_frame(renderer, gpu, use_hiz=True, hiz=hiz, mask=mask, two_pass=True)
    b0 = gpu_frame(renderer, gpu, use_hiz=True, hiz=hiz, mask=mask, two_pass=True, before_pass=camera_b)
    assert_same(a0, a1, KEYS)
    assert_same(b0, b1, KEYS)
    assert a1["share_modes"] == [1, 3] and b1["share_modes"] == [1, 3]
    assert not np.array_equal(a0["late_visible"], b0["late_visible"])


def test_another_call_between_the_two_still_shares_but_the_late_call_prepares_itself(renderer, oracle_lib):
    """early (flag) -> a call of another view on the same context (seeded beforehand, so it neither re-seeds nor rebuilds a list) -> late
    (flag): the late call still matches the early one and reuses its bits, but the accumulators the early call had armed for it are
    no longer taken for granted -- it launches its own prepare kernel.  Bytes as without the flag."""
    from oxylus_amd.renderer import CullGeometryContext, PreparedFrame

    spec = SceneSpec(n_mesh_instances=200, meshlets_per_mesh=500, with_geometry=True, seed=51)
    cpu, gpu, hiz, ohiz, mask = _setup(renderer, spec, 512, 0.3, 51)
    other = make_scene(SceneSpec(n_mesh_instances=50, meshlets_per_mesh=300, with_geometry=True, seed=52), "cpu").to("cuda")
    other_frame = PreparedFrame.create(other, with_triangles=True)
    other_ctx = CullGeometryContext(init_cull_meshes=False, cull_flags=L.CULL_TEST_ALL, cull_camera=other.cull_camera(), stages=L.STAGE_ALL)
    renderer.prepared_frame = other_frame
    renderer.seed_meshlet_instances(other_ctx, other.n_meshlet_instances)
    renderer.cull_geometry(other_ctx)
    want_other = renderer.read_counters(other_ctx).draw_index_count

    def intruder(i, ctx):
        if i == 1:
            mine = renderer.prepared_frame
            renderer.prepared_frame = other_frame
            renderer.cull_geometry(other_ctx)
            assert renderer.read_counters(other_ctx).draw_index_count == want_other
            renderer.prepared_frame = mine

    a = gpu_frame(renderer, gpu, use_hiz=True, hiz=hiz, mask=mask, two_pass=True, share_pass_tests=True, before_pass=intruder)
    b = gpu_frame(renderer, gpu, use_hiz=True, hiz=hiz, mask=mask, two_pass=True)
    assert_same(b, a, KEYS)
    assert a["share_modes"] == [1, 2]
    want = oracle_frame(cpu, use_hiz=True, hiz=ohiz, mask=mask, two_pass=True)
    assert_same(want, a, KEYS)


def test_in_hip_graphs(oracle_lib):
    """(a) The pair captured into one graph and replayed: the late call is part of the same capture as the early call that prepared for it
    (mode 3), every replay starts from re-zeroed accumulators.  (b) Only the late call captured, after an early call outside the graph: it
    reuses the bits (the scene does not change) but must prepare itself (mode 2) -- replayed alone, nobody else would zero its accumulators."""
    from oxylus_amd.renderer import CullGeometryContext, PreparedFrame, RendererInstance

    r = RendererInstance(0)
    spec = SceneSpec(n_mesh_instances=120, meshlets_per_mesh=400, with_geometry=True, seed=61)
    cpu, gpu, hiz, ohiz, mask = _setup(r, spec, 256, 0.3, 61)
    want = oracle_frame(cpu, use_hiz=True, hiz=ohiz, mask=mask, two_pass=True)
    frame = PreparedFrame.create(gpu, with_triangles=True)
    r.prepared_frame = frame
    r.reserve(gpu.n_mesh_instances, gpu.n_meshlet_instances)
    s = torch.cuda.Stream()
    mask_gpu = mask.cuda()

    def check(ctx_e, ctx_l, what):
        torch.cuda.synchronize()
        ce, cl = r.read_counters(ctx_e), r.read_counters(ctx_l)
        assert (cl.early_visible_meshlet_instances, cl.late_visible_meshlet_instances) == (want["early"], want["late"]), what
        vis = frame.visible_meshlet_instances_indices_buffer.cpu().numpy()
        assert np.array_equal(vis[:want["early"]], want["early_visible"]) and np.array_equal(vis[want["early"]:want["early"] + want["late"]], want["late_visible"]), what
        assert cl.draw_index_count == len(want["late_indices"]) and ce.draw_index_count == len(want["early_indices"]), what
        assert np.array_equal(frame.reordered_indices_buffer[:cl.draw_index_count].cpu().numpy(), want["late_indices"]), what
        assert np.array_equal(frame.meshlet_instance_visibility_mask_buffer.cpu().numpy(), want["mask"]), what

    def contexts():
        e = CullGeometryContext(use_hiz=True, init_cull_meshes=False, cull_flags=L.CULL_TEST_ALL, cull_camera=gpu.cull_camera(), hiz_attachment=hiz, stages=L.STAGE_ALL,
                                share_pass_tests=True)
        with torch.cuda.stream(s):
            r.seed_meshlet_instances(e, gpu.n_meshlet_instances, stream=s)
        l = CullGeometryContext(use_hiz=True, init_cull_meshes=False, cull_flags=L.CULL_TEST_ALL | L.CULL_LATE_PASS, cull_camera=gpu.cull_camera(), hiz_attachment=hiz,
                                stages=L.STAGE_ALL, share_pass_tests=True)
        l._c.visibility_buffer, l._c.cull_meshlets_cmd_buffer = e._c.visibility_buffer, e._c.cull_meshlets_cmd_buffer
        return e, l

    # (a) the pair in one graph
    e, l = contexts()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        r.cull_geometry(e, stream=s)
        mode_e = r.debug_shared_tests_mode()
        r.cull_geometry(l, stream=s)
        mode_l = r.debug_shared_tests_mode()
    assert (mode_e, mode_l) == (1, 3)
    for rep in range(3):
        frame.meshlet_instance_visibility_mask_buffer.copy_(mask_gpu)
        frame.visible_meshlet_instances_indices_buffer.zero_()
        torch.cuda.synchronize()
        g.replay()
        check(e, l, f"pair, replay {rep}")
    # (b) the late call alone in a graph
    e, l = contexts()
    frame.meshlet_instance_visibility_mask_buffer.copy_(mask_gpu)
    torch.cuda.synchronize()
    r.cull_geometry(e, stream=s)
    torch.cuda.synchronize()
    mask_after_early = frame.meshlet_instance_visibility_mask_buffer.clone()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=s):
        r.cull_geometry(l, stream=s)
        mode_l = r.debug_shared_tests_mode()
    assert mode_l == 2
    for rep in range(3):
        frame.meshlet_instance_visibility_mask_buffer.copy_(mask_after_early)
        torch.cuda.synchronize()
        g2.replay()
        check(e, l, f"late alone, replay {rep}")
    r.close()


@pytest.mark.parametrize("share", [False, True], ids=["each-call-tests", "share_pass_tests"])
def test_occlusion_candidate_count_is_the_camera_test_survivors(renderer, oracle_lib, share):
    """oxc_debug_count_occlusion_candidates (round 5: SURVEY 8d's f in the bench's rooflines): the counting instantiations of the HiZ meshlet tests
    add the candidates that REACH test_occlusion -- in the late call every meshlet that passes frustum and cone (cull_meshlets_hiz.slang:53-65), in
    the early call those of them that were visible last frame -- and leave every output as it is."""
    import oracle
    from oxylus_amd.renderer import CullGeometryContext, ImageAttachment, MainGeometryContext, PreparedFrame
    from oxylus_amd.synth import make_depth

    spec = SceneSpec(n_mesh_instances=260, meshlets_per_mesh=700, with_geometry=True, seed=77)
    cpu = make_scene(spec, "cpu")
    gpu = cpu.to("cuda")
    hw = 512
    depth = make_depth(2 * hw, 2 * hw, 48, seed=9, device="cuda")
    hiz = ImageAttachment.hiz(hw, hw, "cuda")
    renderer.generate_hiz(MainGeometryContext(ImageAttachment.depth(depth), hiz))
    want_hiz, levels, offs = oracle_hiz(depth.cpu(), hw, hw)
    n = cpu.n_meshlet_instances
    g = torch.Generator().manual_seed(5)
    bits = (torch.rand(((n + 31) // 32, 32), generator=g) < 0.4).to(torch.int64)
    mask = (bits << torch.arange(32)).sum(1).to(torch.int32)
    want = oracle_frame(cpu, use_hiz=True, hiz={"data": want_hiz, "w": hw, "h": hw, "levels": levels, "offs": offs}, mask=mask, two_pass=True)
    # the camera tests alone (frustum + cone) through the plain checker: the late call's candidates; the early call's are those with their mask bit set
    cam_pass = oracle.cull_meshlets(cpu, cpu.cull_camera(), cpu.meshlet_instances).numpy().astype(np.int64)
    was = ((mask.numpy().view(np.uint32)[cam_pass >> 5] >> (cam_pass & 31).astype(np.uint32)) & 1).astype(bool)
    frame = PreparedFrame.create(gpu, with_triangles=True)
    renderer.prepared_frame = frame
    ctx = CullGeometryContext(use_hiz=True, init_cull_meshes=False, cull_flags=L.CULL_TEST_ALL, cull_camera=gpu.cull_camera(), hiz_attachment=hiz, stages=L.STAGE_ALL,
                              share_pass_tests=share)
    renderer.seed_meshlet_instances(ctx, n)
    frame.meshlet_instance_visibility_mask_buffer.copy_(mask.cuda())
    counts = []
    try:
        for tag, flags in (("early", L.CULL_TEST_ALL), ("late", L.CULL_TEST_ALL | L.CULL_LATE_PASS)):
            cnt = torch.zeros(256 * 64, dtype=torch.int32, device="cuda")
            renderer.debug_count_occlusion_candidates(cnt)
            ctx.cull_flags = flags
            renderer.cull_geometry(ctx)
            c = renderer.read_counters(ctx)
            counts.append(int(cnt.to(torch.int64).sum().item()))
            first = c.early_visible_meshlet_instances if tag == "late" else 0
            vis = frame.visible_meshlet_instances_indices_buffer[first:first + c.cull_triangles_cmd_x].cpu().numpy()
            assert np.array_equal(vis, want[f"{tag}_visible"]), tag
            idx = frame.reordered_indices_buffer[:c.draw_index_count].cpu().numpy()
            assert np.array_equal(idx.view(np.uint32), want[f"{tag}_indices"].view(np.uint32)), tag
    finally:
        renderer.debug_count_occlusion_candidates(None)
    assert np.array_equal(frame.meshlet_instance_visibility_mask_buffer.cpu().numpy(), want["mask"])
    assert counts == [int(was.sum()), int(cam_pass.size)], (counts, int(was.sum()), int(cam_pass.size))
    assert 0 < counts[0] < counts[1] < n
