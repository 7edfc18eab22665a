"""GPU: async_triangles (include/oxcull.h) changes WHEN the triangle stage runs, never what it writes.

Several frames are enqueued back to back without a host synchronisation -- every frame with its own camera, so a stale instance
row, a visible list overwritten too early or a counter read too late changes bytes -- once in order on one stream and once with the
triangle stages on the context's own stream; every index list (one buffer per call), the final visible list, the mask and every
call's counters must be identical."""
import ctypes as C
import dataclasses

import numpy as np
import pytest
import torch

from oxylus_amd import lib as L
from oxylus_amd.renderer import CullGeometryContext, ImageAttachment, MainGeometryContext, PreparedFrame, RendererInstance
from oxylus_amd.synth import SceneSpec, make_depth, make_scene

pytestmark = pytest.mark.gpu


def _cameras(scene, n):
    cams = []
    for f in range(n):
        cam = L.CullCamera()
        src = scene.cull_camera()
        C.memmove(C.byref(cam), C.byref(src), C.sizeof(cam))
        sx, sy = 1.0 - 0.07 * f, 1.0 + 0.05 * f  # a different field of view per frame: other planes, other mvp, other triangles
        for c in range(4):
            cam.projection_view[c * 4 + 0] *= sx
            cam.projection_view[c * 4 + 1] *= sy
        cam.position[0] += 0.5 * f
        cams.append(cam)
    return cams


def _run(scene, use_hiz, n_frames, async_mode, mask0, depth, hw, share=False):
    dev = scene.device
    r = RendererInstance(0)
    stream = torch.cuda.Stream(device=dev)
    base = PreparedFrame.create(scene, with_triangles=True)
    base.meshlet_instance_visibility_mask_buffer.copy_(mask0)
    hiz = ImageAttachment.hiz(hw, hw, dev) if use_hiz else None
    passes = [L.CULL_TEST_ALL, L.CULL_TEST_ALL | L.CULL_LATE_PASS] if use_hiz else [L.CULL_TEST_ALL]
    ctx = CullGeometryContext(use_hiz=use_hiz, init_cull_meshes=False, cull_flags=L.CULL_TEST_ALL, cull_camera=scene.cull_camera(), hiz_attachment=hiz,
                              stages=L.STAGE_ALL, async_triangles=async_mode, share_pass_tests=share)
    calls = []
    with torch.cuda.stream(stream):
        r.seed_meshlet_instances(ctx, scene.n_meshlet_instances)
        for cam in _cameras(scene, n_frames):
            if use_hiz:
                r.generate_hiz(MainGeometryContext(ImageAttachment.depth(depth), hiz))
            for flags in passes:
                frame = dataclasses.replace(base, reordered_indices_buffer=torch.zeros_like(base.reordered_indices_buffer))  # own index list, shared everything else
                ctx.cull_flags, ctx.cull_camera = flags, cam
                r.prepared_frame = frame
                r.cull_geometry(ctx)
                snap = L.CullGeometryContext()
                C.memmove(C.byref(snap), C.byref(ctx._c), C.sizeof(snap))
                calls.append((frame, snap))
        r.join_triangles()
    torch.cuda.synchronize()
    out = []
    for frame, snap in calls:
        cnt = L.Counters()
        r._check(r._lib.oxc_read_counters(r._ctx, C.byref(snap), C.byref(cnt), C.c_void_p(stream.cuda_stream)))
        out.append((cnt.cull_triangles_cmd_x, cnt.draw_index_count, frame.reordered_indices_buffer[:cnt.draw_index_count].cpu().numpy().copy()))
    final = (cnt.total_visible_meshlet_instances, cnt.early_visible_meshlet_instances, cnt.late_visible_meshlet_instances,
             base.visible_meshlet_instances_indices_buffer.cpu().numpy().copy(), base.meshlet_instance_visibility_mask_buffer.cpu().numpy().copy())
    r.close()
    return out, final


@pytest.mark.parametrize("use_hiz", [True, False], ids=["hiz-two-pass", "plain"])
def test_async_triangle_stage_writes_the_same_bytes(use_hiz):
    K, M, HW = 1000, 300, 1024
    scene = make_scene(SceneSpec(n_mesh_instances=M, meshlets_per_mesh=K, with_geometry=True, seed=77), "cuda")
    depth = make_depth(2 * HW, 2 * HW, 64, seed=3, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(5)
    words = (scene.n_meshlet_instances + 31) // 32
    bits = (torch.rand((words, 32), generator=g, device="cuda") < 0.3).to(torch.int64)
    mask0 = (bits << torch.arange(32, device="cuda")).sum(1).to(torch.int32)
    want, want_final = _run(scene, use_hiz, 4, False, mask0, depth, HW)
    got, got_final = _run(scene, use_hiz, 4, True, mask0, depth, HW)
    # ... and with the late call of every frame reusing the early call's frustum + cone results (share_pass_tests) on top of it
    got2, got2_final = _run(scene, use_hiz, 4, True, mask0, depth, HW, share=True)
    assert len(want) == len(got) == len(got2)
    for i, (w, g_) in enumerate(zip(want, got2)):
        assert w[0] == g_[0] and w[1] == g_[1] and np.array_equal(w[2], g_[2]), f"call {i} with share_pass_tests"
    assert want_final[:3] == got2_final[:3] and np.array_equal(want_final[3], got2_final[3]) and np.array_equal(want_final[4], got2_final[4])
    assert sum(w[1] for w in want) > 100_000  # the frames emit triangles at all ...
    assert len({w[1] for w in want}) > 2      # ... and differ from each other
    for i, (w, g_) in enumerate(zip(want, got)):
        assert w[0] == g_[0] and w[1] == g_[1], f"call {i}: counters {w[:2]} vs {g_[:2]}"
        assert np.array_equal(w[2], g_[2]), f"call {i}: {int((w[2] != g_[2]).sum())} of {w[2].size} packed indices differ"
    assert want_final[:3] == got_final[:3]
    assert np.array_equal(want_final[3], got_final[3]) and np.array_equal(want_final[4], got_final[4])


def test_in_order_call_after_async_calls_joins_by_itself(renderer):
    """A call without the flag, oxc_read_counters and oxc_pack_counters all wait for what is in flight on the side stream."""
    scene = make_scene(SceneSpec(n_mesh_instances=64, meshlets_per_mesh=500, with_geometry=True, seed=9), "cuda")
    frame = PreparedFrame.create(scene, with_triangles=True)
    renderer.prepared_frame = frame
    ctx = CullGeometryContext(init_cull_meshes=False, cull_flags=L.CULL_TEST_ALL, cull_camera=scene.cull_camera(), stages=L.STAGE_ALL)
    renderer.seed_meshlet_instances(ctx, scene.n_meshlet_instances)
    renderer.cull_geometry(ctx)
    c0 = renderer.read_counters(ctx)
    want = frame.reordered_indices_buffer[:c0.draw_index_count].clone()
    for _ in range(3):  # async, async, then in order: the last one must find the scratch free and leave the same list
        frame.reordered_indices_buffer.zero_()
        ctx.async_triangles = True
        renderer.cull_geometry(ctx)
        renderer.cull_geometry(ctx)
        ctx.async_triangles = False
        renderer.cull_geometry(ctx)
        c = renderer.read_counters(ctx)
        assert c.draw_index_count == c0.draw_index_count
        assert torch.equal(frame.reordered_indices_buffer[:c.draw_index_count], want)
    ctx.async_triangles = True
    frame.reordered_indices_buffer.zero_()
    renderer.cull_geometry(ctx)
    c = renderer.read_counters(ctx)  # joins
    assert c.draw_index_count == c0.draw_index_count and torch.equal(frame.reordered_indices_buffer[:c.draw_index_count], want)


def test_async_pair_captured_into_a_hip_graph_after_an_eager_async_warm_up(oracle_lib):
    """Round-3 advisor finding: the pending-stage bookkeeping did not know about stream capture.  An eager async call (needed before any
    capture: scratch is allocated un-captured) left a pending entry whose event was recorded OUTSIDE the capture; the first captured call
    then waited for it -> hipErrorStreamCaptureIsolation, capture invalidated.  Now: eager async warm-up (joined), then the async early +
    late pair and the join captured into one graph, replayed three times; afterwards un-captured async calls again (their bookkeeping must
    not wait for events recorded inside the finished capture).  Every replay leaves the bytes of the in-order frame."""
    from util import gpu_frame, oracle_frame, oracle_hiz

    r = RendererInstance(0)
    spec = SceneSpec(n_mesh_instances=120, meshlets_per_mesh=400, with_geometry=True, seed=61)
    cpu = make_scene(spec, "cpu")
    gpu = cpu.to("cuda")
    hw = 256
    depth = make_depth(2 * hw, 2 * hw, 48, seed=61, device="cuda")
    hiz = ImageAttachment.hiz(hw, hw, "cuda")
    r.generate_hiz(MainGeometryContext(ImageAttachment.depth(depth), hiz))
    want_hiz, levels, offs = oracle_hiz(depth.cpu(), hw, hw)
    n = cpu.n_meshlet_instances
    g = torch.Generator().manual_seed(61)
    bits = (torch.rand(((n + 31) // 32, 32), generator=g) < 0.3).to(torch.int64)
    mask = (bits << torch.arange(32)).sum(1).to(torch.int32)
    want = oracle_frame(cpu, use_hiz=True, hiz={"data": want_hiz, "w": hw, "h": hw, "levels": levels, "offs": offs}, mask=mask, two_pass=True)
    frame = PreparedFrame.create(gpu, with_triangles=True)
    r.prepared_frame = frame
    r.reserve(gpu.n_mesh_instances, n)
    s = torch.cuda.Stream()
    mask_gpu = mask.cuda()

    def contexts():
        e = CullGeometryContext(use_hiz=True, init_cull_meshes=False, cull_flags=L.CULL_TEST_ALL, cull_camera=gpu.cull_camera(), hiz_attachment=hiz, stages=L.STAGE_ALL,
                                async_triangles=True)
        with torch.cuda.stream(s):
            r.seed_meshlet_instances(e, n, stream=s)
        l = CullGeometryContext(use_hiz=True, init_cull_meshes=False, cull_flags=L.CULL_TEST_ALL | L.CULL_LATE_PASS, cull_camera=gpu.cull_camera(), hiz_attachment=hiz,
                                stages=L.STAGE_ALL, async_triangles=True)
        l._c.visibility_buffer, l._c.cull_meshlets_cmd_buffer = e._c.visibility_buffer, e._c.cull_meshlets_cmd_buffer
        return e, l

    def check(ctx_e, ctx_l, what):
        torch.cuda.synchronize()
        ce, cl = r.read_counters(ctx_e), r.read_counters(ctx_l)
        assert (cl.early_visible_meshlet_instances, cl.late_visible_meshlet_instances) == (want["early"], want["late"]), what
        vis = frame.visible_meshlet_instances_indices_buffer.cpu().numpy()
        assert np.array_equal(vis[:want["early"]], want["early_visible"]) and np.array_equal(vis[want["early"]:want["early"] + want["late"]], want["late_visible"]), what
        assert cl.draw_index_count == len(want["late_indices"]) and ce.draw_index_count == len(want["early_indices"]), what
        assert np.array_equal(frame.reordered_indices_buffer[:cl.draw_index_count].cpu().numpy(), want["late_indices"]), what
        assert np.array_equal(frame.meshlet_instance_visibility_mask_buffer.cpu().numpy(), want["mask"]), what

    # eager async warm-up, joined on the stream and complete on the host before the capture begins (include/oxcull.h)
    e, l = contexts()
    frame.meshlet_instance_visibility_mask_buffer.copy_(mask_gpu)
    torch.cuda.synchronize()
    r.cull_geometry(e, stream=s)
    r.cull_geometry(l, stream=s)
    r.join_triangles(stream=s)
    check(e, l, "eager async warm-up")
    # the pair + the join in one graph
    e, l = contexts()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        r.cull_geometry(e, stream=s)
        r.cull_geometry(l, stream=s)
        r.join_triangles(stream=s)
    for rep in range(3):
        frame.meshlet_instance_visibility_mask_buffer.copy_(mask_gpu)
        frame.visible_meshlet_instances_indices_buffer.zero_()
        frame.reordered_indices_buffer.zero_()
        torch.cuda.synchronize()
        graph.replay()
        check(e, l, f"captured async pair, replay {rep}")
    # ... and un-captured async calls afterwards
    e2, l2 = contexts()
    frame.meshlet_instance_visibility_mask_buffer.copy_(mask_gpu)
    torch.cuda.synchronize()
    r.cull_geometry(e2, stream=s)
    r.cull_geometry(l2, stream=s)
    r.join_triangles(stream=s)
    check(e2, l2, "eager async after the capture")
    r.close()
