"""GPU parity, round 2: the BASELINE configurations that had no HIP-path test (configs[2] at full size with the triangle stage,
configs[4] multi-view with per-view LOD), the opt-in small-triangle cull, and the robustness rules the ABI states (calls of one
context on different streams are ordered, scratch never grows inside a stream capture, device-side list lengths are clamped
to the caller's buffers)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import oracle
from oxylus_amd import lib as L
from oxylus_amd.renderer import CullGeometryContext, ImageAttachment, MainGeometryContext, PreparedFrame, RendererInstance
from oxylus_amd.synth import SceneSpec, make_depth, make_scene, virtual_shadow_matrices

from util import assert_same, gpu_frame, oracle_frame, oracle_hiz

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


# ------------------------------------------------------------------------------------------------------------------
# configs[2]: the whole north-star path at FULL size against the checker, every output, bit for bit
# ------------------------------------------------------------------------------------------------------------------
def test_config3_full_size_every_output_bit_exact(renderer, oracle_lib):
    """10M meshlets with unique geometry (~10 GB) + 4096^2 HiZ from an 8192^2 depth + prior mask p = 0.3: HiZ build, early
    and late meshlet passes, both triangle passes.  The checker runs the same sequence over the WHOLE scene (host copy); the
    pyramid, both visible lists (1.08M meshlets), both packed index lists (35M triangles), the mask and every counter must be
    byte-identical, unsorted."""
    K, M, HW = 1000, 10_000, 4096
    gpu = make_scene(SceneSpec(n_mesh_instances=M, meshlets_per_mesh=K, with_geometry=True, seed=0x0A1DE5 + 2), "cuda")
    N = gpu.n_meshlet_instances
    depth = make_depth(2 * HW, 2 * HW, 64, seed=3, device="cuda")
    hiz = ImageAttachment.hiz(HW, HW, "cuda")
    g = torch.Generator(device="cuda").manual_seed(5)
    words = (N + 31) // 32
    bits = (torch.rand((words, 32), generator=g, device="cuda") < 0.3).to(torch.int64)
    mask0 = (bits << torch.arange(32, device="cuda")).sum(1).to(torch.int32)
    del bits
    frame = PreparedFrame.create(gpu, with_triangles=True)
    frame.meshlet_instance_visibility_mask_buffer.copy_(mask0)
    renderer.prepared_frame = frame
    renderer.generate_hiz(MainGeometryContext(ImageAttachment.depth(depth), hiz))
    runs = {}
    for share in (False, True):  # every call on its own / the late call reusing the early call's frustum + cone results (share_pass_tests)
        frame.meshlet_instance_visibility_mask_buffer.copy_(mask0)
        frame.visible_meshlet_instances_indices_buffer.zero_()
        ctx = CullGeometryContext(use_hiz=True, init_cull_meshes=False, cull_flags=L.CULL_TEST_ALL, cull_camera=gpu.cull_camera(), hiz_attachment=hiz, stages=L.STAGE_ALL,
                                  share_pass_tests=share)
        renderer.seed_meshlet_instances(ctx, N)
        got = {}
        for tag, flags in (("early", L.CULL_TEST_ALL), ("late", L.CULL_TEST_ALL | L.CULL_LATE_PASS)):
            ctx.cull_flags = flags
            renderer.cull_geometry(ctx)
            c = renderer.read_counters(ctx)
            first = c.early_visible_meshlet_instances if tag == "late" else 0
            got[tag] = (frame.visible_meshlet_instances_indices_buffer[first:first + c.cull_triangles_cmd_x].cpu(), frame.reordered_indices_buffer[:c.draw_index_count].cpu(),
                        (c.total_visible_meshlet_instances, c.early_visible_meshlet_instances, c.late_visible_meshlet_instances, c.cull_triangles_cmd_x))
        got["mask"] = frame.meshlet_instance_visibility_mask_buffer.cpu()
        runs[share] = got
    for tag in ("early", "late"):
        assert runs[True][tag][2] == runs[False][tag][2], f"{tag}: counters differ with share_pass_tests"
        assert torch.equal(runs[True][tag][0], runs[False][tag][0]) and torch.equal(runs[True][tag][1], runs[False][tag][1]), f"{tag}: lists differ with share_pass_tests"
    assert torch.equal(runs[True]["mask"], runs[False]["mask"])
    # ... and unordered_output = 1 (the reference's atomic slot allocation in the triangle stage, one launch; the triangle kernel finds its ids itself): same
    # counters, same mask, and every list SORTED is the ordered list (which is compared with the checker below)
    for mode in (1,):
        frame.meshlet_instance_visibility_mask_buffer.copy_(mask0)
        frame.visible_meshlet_instances_indices_buffer.zero_()
        ctx = CullGeometryContext(use_hiz=True, init_cull_meshes=False, cull_flags=L.CULL_TEST_ALL, cull_camera=gpu.cull_camera(), hiz_attachment=hiz, stages=L.STAGE_ALL,
                                  share_pass_tests=True, unordered_output=mode)
        renderer.seed_meshlet_instances(ctx, N)
        for tag, flags in (("early", L.CULL_TEST_ALL), ("late", L.CULL_TEST_ALL | L.CULL_LATE_PASS)):
            ctx.cull_flags = flags
            renderer.cull_geometry(ctx)
            c = renderer.read_counters(ctx)
            first = c.early_visible_meshlet_instances if tag == "late" else 0
            assert (c.total_visible_meshlet_instances, c.early_visible_meshlet_instances, c.late_visible_meshlet_instances, c.cull_triangles_cmd_x) == runs[False][tag][2], (mode, tag)
            assert c.draw_index_count == runs[False][tag][1].numel(), (mode, tag)
            vis = torch.sort(frame.visible_meshlet_instances_indices_buffer[first:first + c.cull_triangles_cmd_x])[0].cpu()
            assert torch.equal(vis, runs[False][tag][0]), f"unordered_output = {mode}, {tag}: visible set differs"
            idx = torch.sort(frame.reordered_indices_buffer[:c.draw_index_count].to(torch.int64) & 0xFFFFFFFF)[0].to(torch.int32).cpu()
            assert torch.equal(idx, runs[False][tag][1]), f"unordered_output = {mode}, {tag}: packed index set differs"
            del vis, idx
        assert torch.equal(frame.meshlet_instance_visibility_mask_buffer.cpu(), runs[False]["mask"]), f"unordered_output = {mode}: mask differs"
    got, got_mask = runs[False], runs[False]["mask"]
    del runs
    got_hiz = hiz.data.cpu()
    # ---- the checker, whole scene
    cpu = gpu.to("cpu")
    del gpu, frame
    torch.cuda.empty_cache()
    want_hiz, levels, offs = oracle_hiz(depth.cpu(), HW, HW)
    assert torch.equal(want_hiz.view(torch.int32), got_hiz.view(torch.int32)), "pyramid differs"
    hz = oracle.make_hiz(want_hiz, HW, HW, levels, offs)
    cam = cpu.cull_camera()
    v = oracle.Visibility(N, 0, 0)
    out = torch.zeros(N, dtype=torch.int32)
    mask = mask0.cpu().clone()
    for tag, flags in (("early", L.CULL_TEST_ALL), ("late", L.CULL_TEST_ALL | L.CULL_LATE_PASS)):
        n_e = oracle.cull_meshlets_hiz(cpu, cam, cpu.meshlet_instances, flags, hz, v, mask, out)
        first = v.early if tag == "late" else 0
        want_vis = out[first:first + n_e]
        assert got[tag][2] == (N, v.early, v.late, n_e), (tag, got[tag][2], (N, v.early, v.late, n_e))
        assert torch.equal(got[tag][0], want_vis), f"{tag}: visible list differs"
        want_idx = oracle.cull_triangles(cpu, cam, cpu.meshlet_instances, out, first, n_e, nthreads=16)
        assert torch.equal(got[tag][1], want_idx), f"{tag}: packed triangle indices differ ({got[tag][1].numel()} vs {want_idx.numel()})"
    assert torch.equal(got_mask, mask)
    assert v.early + v.late > 1_000_000  # more than a million visible meshlets went through the triangle stage


# ------------------------------------------------------------------------------------------------------------------
# configs[4]: 16 orthographic cascade views, per-view cull_meshes (frustum + LOD select), one batched call
# ------------------------------------------------------------------------------------------------------------------
def _cascade_cameras(scene, views):
    mats, _, zn = virtual_shadow_matrices([0.0, 0.0, -60.0], [0.3, -1.0, 0.2], 500.0, 2.0, views)  # Shadowmaps.cpp:9-63, doubling extents
    cams = []
    for v in range(views):
        cam = scene.cull_camera()
        for k in range(16):
            cam.projection_view[k] = float(mats[v][k])
        cam.position[0], cam.position[1], cam.position[2] = 0.0, 0.0, -60.0
        cam.near_clip = zn
        cams.append(cam)
    return cams


def _run_views_batched(renderer, scene, cams, flags):
    import dataclasses

    frames = [PreparedFrame.create(scene if e == 0 else dataclasses.replace(scene, mesh_instances=scene.mesh_instances.clone()), with_triangles=False, expand=False)
              for e in range(len(cams))]
    ctxs = [CullGeometryContext(init_cull_meshes=True, cull_flags=flags, cull_camera=cam, stages=L.STAGE_MESHES | L.STAGE_MESHLETS) for cam in cams]
    renderer.cull_geometry_batch(frames, ctxs)
    out = []
    for f, c in zip(frames, ctxs):
        cnt = renderer.read_counters(c)
        out.append({"total": cnt.total_visible_meshlet_instances, "visible": f.visible_meshlet_instances_indices_buffer[:cnt.cull_triangles_cmd_x].cpu(),
                    "mli": f.meshlet_instances_buffer[:cnt.total_visible_meshlet_instances].cpu(), "lod": f.scene.mesh_instances[:, 1].cpu()})
    return out


def test_config5_cascade_views_per_view_lod_vs_oracle(renderer, oracle_lib):
    """16 orthographic cascades over a 4-LOD scene: per view the expanded MeshletInstance list (LOD-selected), the lod_index
    column and the visible list must equal the checker's; the views must differ from each other (several LODs, different counts)."""
    views = 16
    cpu = make_scene(SceneSpec(n_mesh_instances=1500, meshlets_per_mesh=96, lod_count=4, seed=0x0A1DE5 + 4, with_geometry=False), "cpu")
    gpu = cpu.to("cuda")
    flags = L.CULL_TEST_FRUSTUM | L.CULL_SELECT_LOD
    got = _run_views_batched(renderer, gpu, _cascade_cameras(gpu, views), flags)
    totals, lods_seen = [], set()
    for v, cam in enumerate(_cascade_cameras(cpu, views)):
        s = cpu.clone()  # cull_meshes writes lod_index
        mli, _ = oracle.cull_meshes(s, cam, flags)
        vis = oracle.cull_meshlets(s, cam, mli)
        assert got[v]["total"] == mli.shape[0], f"view {v}"
        assert torch.equal(got[v]["mli"], mli), f"view {v}: expansion differs"
        assert torch.equal(got[v]["visible"], vis), f"view {v}: visible list differs"
        emitted = torch.unique(mli[:, 0].to(torch.int64))
        assert torch.equal(got[v]["lod"][emitted], s.mesh_instances[emitted, 1]), f"view {v}: lod_index differs"
        totals.append(mli.shape[0])
        lods_seen |= set(s.mesh_instances[emitted, 1].tolist())
    assert len(set(totals)) > 4 and len(lods_seen) > 1, (totals, lods_seen)


def test_config5_full_size_properties(renderer, oracle_lib):
    """configs[4] at full size (10M LOD-0 meshlets x 16 views, one batched call): per view the list is ascending, counts are
    consistent, the expansion is complete (every record is (instance, 0..count-1) in order) and the batched call equals single
    calls for two of the views."""
    views = 16
    gpu = make_scene(SceneSpec(n_mesh_instances=10_000, meshlets_per_mesh=1000, lod_count=3, seed=0x0A1DE5 + 4, with_geometry=False), "cuda")
    flags = L.CULL_TEST_FRUSTUM | L.CULL_SELECT_LOD
    cams = _cascade_cameras(gpu, views)
    got = _run_views_batched(renderer, gpu, cams, flags)
    lod_counts = torch.tensor(gpu.lod_meshlet_counts)
    for v in range(views):
        r = got[v]
        vis, mli = r["visible"], r["mli"]
        assert r["total"] == mli.shape[0] and vis.numel() <= r["total"] <= gpu.n_meshlet_instances
        if vis.numel() > 1:
            assert bool((vis[1:] > vis[:-1]).all()) and int(vis[-1]) < r["total"]
        if mli.shape[0]:
            inst, k = mli[:, 0].to(torch.int64), mli[:, 1].to(torch.int64)
            assert bool((inst[1:] >= inst[:-1]).all())
            start = torch.ones_like(inst, dtype=torch.bool)
            start[1:] = inst[1:] != inst[:-1]
            assert bool((k[start] == 0).all()) and bool((k[~start] == k[torch.nonzero(~start).squeeze(1) - 1] + 1).all())
            uniq, cnt = torch.unique_consecutive(inst, return_counts=True)
            assert torch.equal(cnt, lod_counts[r["lod"][uniq].to(torch.int64)]), "every emitted instance carries all meshlets of its selected LOD"
    assert got[0]["total"] < got[views - 1]["total"]  # the smallest cascade sees fewer instances than the largest
    for v in (0, views - 1):
        frame = PreparedFrame.create(gpu, with_triangles=False, expand=False)
        renderer.prepared_frame = frame
        ctx = CullGeometryContext(init_cull_meshes=True, cull_flags=flags, cull_camera=cams[v], stages=L.STAGE_MESHES | L.STAGE_MESHLETS)
        renderer.cull_geometry(ctx)
        c = renderer.read_counters(ctx)
        assert c.total_visible_meshlet_instances == got[v]["total"]
        assert torch.equal(frame.visible_meshlet_instances_indices_buffer[:c.cull_triangles_cmd_x].cpu(), got[v]["visible"])


@pytest.mark.parametrize("views,move_cameras,cap_frac", [(2, False, 1.0), (5, True, 1.0), (16, True, 1.0), (7, False, 0.4), (6, "rotate", 1.0), (9, "mixed", 1.0)],
                         ids=["2-views", "5-views-own-positions", "16-views-own-positions", "7-views-short-lists", "6-views-one-position-own-orientations",
                              "9-views-one-position-cascades-and-others"])
def test_multiview_batch_equals_single_calls(renderer, oracle_lib, views, move_cameras, cap_frac):
    """The batched views of one scene take the one-pass multi-view meshlet stage; every element's outputs -- the LOD-selected
    MeshletInstance list, the visible list, the counters -- must equal what a single oxc_cull_geometry call of that view writes.  Own camera
    positions per view (the normal cone is then evaluated per view), a view that sees nothing, and lists cut short by the caller's
    buffers (max_meshlet_instance_count smaller than what cull_meshes emits) included."""
    import dataclasses

    gpu = make_scene(SceneSpec(n_mesh_instances=700, meshlets_per_mesh=150, lod_count=3, seed=0x0A1DE5 + 9, with_geometry=False), "cuda")
    flags = L.CULL_TEST_FRUSTUM | L.CULL_SELECT_LOD
    cams = _cascade_cameras(gpu, views)
    if move_cameras in ("rotate", "mixed"):
        # One camera position, plane normals that are NOT the leader's (round 4: the kernel shares a box's plane distances between the views whose
        # normals are, k_mv_group's mask; the others take their own frustum test inside the same step): perspective views turned about the y axis
        # ("mixed": every third view; the rest stay cascades of the one light).
        base = np.array(gpu.cull_camera().projection_view, dtype=np.float64).reshape(4, 4).T  # column-major -> matrix
        for v, cam in enumerate(cams):
            if move_cameras == "mixed" and v % 3 != 1:
                continue
            t = 0.35 * (v + 1)
            rot = np.array([[np.cos(t), 0, np.sin(t), 0], [0, 1, 0, 0], [-np.sin(t), 0, np.cos(t), 0], [0, 0, 0, 1]])
            m = (base @ rot).T.reshape(-1).astype(np.float32)
            for k in range(16):
                cam.projection_view[k] = float(m[k])
            cam.near_clip = gpu.cull_camera().near_clip
    elif move_cameras:
        for v, cam in enumerate(cams):
            cam.position[0], cam.position[1], cam.position[2] = 3.0 * v, -2.0 * v, -60.0 + 11.0 * v
        for k in range(16):  # the last view looks away from everything: no instance survives its cull_meshes
            cams[-1].projection_view[k] = float(gpu.cull_camera().projection_view[k]) * (-1.0 if k % 4 == 2 else 1.0)
    cap = max(64, int(gpu.n_meshlet_instances * cap_frac))

    def frame_of(e):
        f = PreparedFrame.create(gpu if e == 0 else dataclasses.replace(gpu, mesh_instances=gpu.mesh_instances.clone()), with_triangles=False, expand=False)
        if cap_frac < 1.0:
            f.max_meshlet_instance_count = cap
        return f

    frames = [frame_of(e) for e in range(views)]
    ctxs = [CullGeometryContext(init_cull_meshes=True, cull_flags=flags, cull_camera=cam, stages=L.STAGE_MESHES | L.STAGE_MESHLETS) for cam in cams]
    renderer.cull_geometry_batch(frames, ctxs)
    nonempty = 0
    for v in range(views):
        got = renderer.read_counters(ctxs[v])
        single = frame_of(1)
        renderer.prepared_frame = single
        c1 = CullGeometryContext(init_cull_meshes=True, cull_flags=flags, cull_camera=cams[v], stages=L.STAGE_MESHES | L.STAGE_MESHLETS)
        renderer.cull_geometry(c1)
        want = renderer.read_counters(c1)
        assert (got.total_visible_meshlet_instances, got.cull_triangles_cmd_x) == (want.total_visible_meshlet_instances, want.cull_triangles_cmd_x), f"view {v}"
        n, e = want.total_visible_meshlet_instances, want.cull_triangles_cmd_x
        assert torch.equal(frames[v].meshlet_instances_buffer[:n], single.meshlet_instances_buffer[:n]), f"view {v}: expansion differs"
        assert torch.equal(frames[v].visible_meshlet_instances_indices_buffer[:e], single.visible_meshlet_instances_indices_buffer[:e]), f"view {v}: visible list differs"
        nonempty += int(e > 0)
        if cap_frac < 1.0:
            assert n <= cap
    assert nonempty >= 2


@pytest.mark.parametrize("views,cap_frac", [(5, 1.0), (16, 1.0), (7, 0.4)], ids=["5-views", "16-views", "7-views-short-lists"])
def test_multiview_batch_with_implicit_meshlet_instance_lists(renderer, oracle_lib, views, cap_frac):
    """implicit_meshlet_instances (include/oxcull.h): the per-view MeshletInstance records are not written -- {first, count} runs per mesh
    instance are -- and nothing else changes: counters, lod_index and visible lists are those of the explicit batch, the record buffers
    stay untouched, and expanding the runs gives the explicit list byte for byte.  Also: the runs buffer alone (flag off), and the calls
    that must refuse the flag."""
    import dataclasses

    gpu = make_scene(SceneSpec(n_mesh_instances=700, meshlets_per_mesh=150, lod_count=3, seed=0x0A1DE5 + 9, with_geometry=False), "cuda")
    M = gpu.n_mesh_instances
    flags = L.CULL_TEST_FRUSTUM | L.CULL_SELECT_LOD
    cams = _cascade_cameras(gpu, views)
    for v, cam in enumerate(cams):
        cam.position[0], cam.position[1], cam.position[2] = 3.0 * v, -2.0 * v, -60.0 + 11.0 * v
    cap = max(64, int(gpu.n_meshlet_instances * cap_frac))

    def frame_of(e):
        f = PreparedFrame.create(gpu if e == 0 else dataclasses.replace(gpu, mesh_instances=gpu.mesh_instances.clone()), with_triangles=False, expand=False)
        f.meshlet_instances_buffer.fill_(-7)
        if cap_frac < 1.0:
            f.max_meshlet_instance_count = cap
        return f

    def run(implicit, with_runs):
        frames = [frame_of(e) for e in range(views)]
        runs = [torch.full((M, 2), -1, dtype=torch.int32, device="cuda") if with_runs else None for _ in range(views)]
        ctxs = [CullGeometryContext(init_cull_meshes=True, cull_flags=flags, cull_camera=cam, stages=L.STAGE_MESHES | L.STAGE_MESHLETS,
                                    implicit_meshlet_instances=implicit, meshlet_instance_runs_buffer=runs[v]) for v, cam in enumerate(cams)]
        renderer.cull_geometry_batch(frames, ctxs)
        return frames, ctxs, runs, [renderer.read_counters(c) for c in ctxs]

    f0, c0, _, n0 = run(False, False)
    f1, c1, r1, n1 = run(True, True)
    f2, c2, r2, n2 = run(False, True)
    total = 0
    for v in range(views):
        a, b = n0[v], n1[v]
        assert (a.total_visible_meshlet_instances, a.cull_triangles_cmd_x, a.cull_meshlets_cmd_x) == (b.total_visible_meshlet_instances, b.cull_triangles_cmd_x, b.cull_meshlets_cmd_x), f"view {v}"
        n, e = a.total_visible_meshlet_instances, a.cull_triangles_cmd_x
        total += n
        assert torch.equal(f0[v].visible_meshlet_instances_indices_buffer[:e], f1[v].visible_meshlet_instances_indices_buffer[:e]), f"view {v}: visible list"
        assert torch.equal(f0[v].scene.mesh_instances[:, 1], f1[v].scene.mesh_instances[:, 1]), f"view {v}: lod_index"
        assert int((f1[v].meshlet_instances_buffer != -7).sum()) == 0, f"view {v}: records were written although the list is implicit"
        # the runs, expanded, ARE the explicit list
        first, count = r1[v][:, 0].long(), r1[v][:, 1].long()
        assert int(count.sum()) == n and torch.equal(r1[v], r2[v])
        inst = torch.repeat_interleave(torch.arange(M, device="cuda"), count)
        start = torch.repeat_interleave(first, count)
        want = torch.stack([inst, torch.arange(n, device="cuda") - start], 1).to(torch.int32)
        kept = count > 0
        assert torch.equal(first[kept], (torch.cumsum(count, 0) - count)[kept]), f"view {v}: runs are not the list's prefix sums"
        assert torch.equal(want, f0[v].meshlet_instances_buffer[:n]), f"view {v}: expanded runs differ from the explicit list"
        assert torch.equal(f2[v].meshlet_instances_buffer[:n], f0[v].meshlet_instances_buffer[:n])
    assert total > 10_000
    # refused: a single call, a batch in which only some elements set it, a batch with the triangle stage
    renderer.prepared_frame = frame_of(1)
    bad = CullGeometryContext(init_cull_meshes=True, cull_flags=flags, cull_camera=cams[0], stages=L.STAGE_MESHES | L.STAGE_MESHLETS, implicit_meshlet_instances=True,
                              meshlet_instance_runs_buffer=torch.zeros((M, 2), dtype=torch.int32, device="cuda"))
    with pytest.raises(L.OxcError) as ei:
        renderer.cull_geometry(bad)
    assert ei.value.status == L.OXC_INVALID_ARG
    frames = [frame_of(e) for e in range(2)]
    mixed = [CullGeometryContext(init_cull_meshes=True, cull_flags=flags, cull_camera=cams[v], stages=L.STAGE_MESHES | L.STAGE_MESHLETS, implicit_meshlet_instances=(v == 0),
                                 meshlet_instance_runs_buffer=torch.zeros((M, 2), dtype=torch.int32, device="cuda")) for v in range(2)]
    with pytest.raises(L.OxcError) as ei:
        renderer.cull_geometry_batch(frames, mixed)
    assert ei.value.status == L.OXC_INVALID_ARG


def test_pack_counters_batch_matches_read_counters(renderer, oracle_lib):
    import ctypes as C
    import dataclasses

    gpu = make_scene(SceneSpec(n_mesh_instances=300, meshlets_per_mesh=120, lod_count=2, seed=0x0A1DE5 + 12, with_geometry=False), "cuda")
    cams = _cascade_cameras(gpu, 6)
    frames = [PreparedFrame.create(gpu if e == 0 else dataclasses.replace(gpu, mesh_instances=gpu.mesh_instances.clone()), with_triangles=False, expand=False) for e in range(6)]
    ctxs = [CullGeometryContext(init_cull_meshes=True, cull_flags=L.CULL_TEST_FRUSTUM | L.CULL_SELECT_LOD, cull_camera=cam, stages=L.STAGE_MESHES | L.STAGE_MESHLETS) for cam in cams]
    renderer.cull_geometry_batch(frames, ctxs)
    cc = (L.CullGeometryContext * 6)(*[c._c for c in ctxs])
    out = torch.full((6, 4), -1, dtype=torch.int32, device="cuda")
    renderer._check(renderer._lib.oxc_pack_counters_batch(renderer._ctx, 6, cc, C.c_void_p(out.data_ptr()), renderer._stream(None)))
    torch.cuda.synchronize()
    for v in range(6):
        cnt = renderer.read_counters(ctxs[v])
        assert out[v].tolist() == [cnt.cull_triangles_cmd_x, cnt.total_visible_meshlet_instances, cnt.late_visible_meshlet_instances, cnt.draw_index_count]
    assert int(out[:, 0].sum()) > 0


def test_multiview_batch_with_the_triangle_stage(renderer, oracle_lib):
    """Three views of one scene with every stage: the per-view triangle kernels consume the visible lists the multi-view meshlet stage
    wrote; packed indices and draw commands must equal those of single calls."""
    import dataclasses

    gpu = make_scene(SceneSpec(n_mesh_instances=300, meshlets_per_mesh=120, lod_count=2, seed=0x0A1DE5 + 11, with_geometry=True), "cuda")
    flags = L.CULL_TEST_FRUSTUM | L.CULL_SELECT_LOD
    cams = _cascade_cameras(gpu, 6)[3:]

    def frame_of(e):
        return PreparedFrame.create(gpu if e == 0 else dataclasses.replace(gpu, mesh_instances=gpu.mesh_instances.clone()), with_triangles=True, expand=False)

    frames = [frame_of(e) for e in range(3)]
    ctxs = [CullGeometryContext(init_cull_meshes=True, cull_flags=flags, cull_camera=cam, stages=L.STAGE_ALL) for cam in cams]
    renderer.cull_geometry_batch(frames, ctxs)
    emitted = 0
    for v in range(3):
        got = renderer.read_counters(ctxs[v])
        single = frame_of(1)
        renderer.prepared_frame = single
        c1 = CullGeometryContext(init_cull_meshes=True, cull_flags=flags, cull_camera=cams[v], stages=L.STAGE_ALL)
        renderer.cull_geometry(c1)
        want = renderer.read_counters(c1)
        assert (got.cull_triangles_cmd_x, got.draw_index_count) == (want.cull_triangles_cmd_x, want.draw_index_count), f"view {v}"
        assert torch.equal(frames[v].reordered_indices_buffer[:want.draw_index_count], single.reordered_indices_buffer[:want.draw_index_count]), f"view {v}: packed indices differ"
        emitted += want.draw_index_count
    assert emitted > 10_000


# ------------------------------------------------------------------------------------------------------------------
# opt-in small-triangle cull (include/oxcull.h: oxc_cull_geometry_context::small_triangle_cull)
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("spec,res", [(SceneSpec(n_mesh_instances=40, meshlets_per_mesh=120, seed=31), 256), (SceneSpec(n_mesh_instances=9, meshlets_per_mesh=300, seed=33, ragged=True), 64),
                                      (SceneSpec(n_mesh_instances=16, meshlets_per_mesh=64, seed=35, nonuniform_scale=True), 4096)], ids=["res256", "ragged-res64", "res4096"])
def test_small_triangle_cull_on_and_off(renderer, oracle_lib, spec, res):
    spec.resolution = res
    cpu = make_scene(spec, "cpu")
    gpu = cpu.to("cuda")
    cam = cpu.cull_camera()
    vis = oracle.cull_meshlets(cpu, cam, cpu.meshlet_instances)
    want_off = oracle.cull_triangles(cpu, cam, cpu.meshlet_instances, vis, 0, vis.numel())
    want_on = oracle.cull_triangles(cpu, cam, cpu.meshlet_instances, vis, 0, vis.numel(), small_triangle_cull=True)
    res_ = {}
    for on in (False, True):
        frame = PreparedFrame.create(gpu)
        renderer.prepared_frame = frame
        ctx = CullGeometryContext(init_cull_meshes=False, cull_flags=L.CULL_TEST_ALL, cull_camera=gpu.cull_camera(), small_triangle_cull=on)
        renderer.seed_meshlet_instances(ctx, gpu.n_meshlet_instances)
        renderer.cull_geometry(ctx)
        c = renderer.read_counters(ctx)
        res_[on] = frame.reordered_indices_buffer[:c.draw_index_count].cpu()
    assert torch.equal(res_[False], want_off), "flag off must be the reference behaviour"
    assert torch.equal(res_[True], want_on)
    assert want_on.numel() < want_off.numel() or res == 4096
    # what the flag drops is a subset of what the reference keeps
    assert set(want_on.view(-1, 3)[:, 0].tolist()) <= set(want_off.view(-1, 3)[:, 0].tolist())


def test_small_triangle_flag_off_matches_the_round1_fixture(renderer, oracle_lib):
    """Byte-identical to the committed round-1 fixture (generated before the flag existed) with the flag explicitly off."""
    from util import scene_from_golden

    s, z = scene_from_golden(os.path.join(HERE, "golden", "pipeline_12x40.npz"), "cuda")
    frame = PreparedFrame.create(s, expand=False)
    renderer.prepared_frame = frame
    ctx = CullGeometryContext(init_cull_meshes=True, cull_flags=L.CULL_TEST_ALL, cull_camera=s.cull_camera(), small_triangle_cull=False)
    renderer.cull_geometry(ctx)
    c = renderer.read_counters(ctx)
    got = frame.reordered_indices_buffer[:c.draw_index_count].cpu().numpy()
    assert np.array_equal(got.view(np.uint32), np.asarray(z["plain_indices"]).view(np.uint32))
    ctx2 = CullGeometryContext(init_cull_meshes=True, cull_flags=L.CULL_TEST_ALL, cull_camera=s.cull_camera(), small_triangle_cull=True)
    renderer.cull_geometry(ctx2)
    assert renderer.read_counters(ctx2).draw_index_count <= c.draw_index_count


# ------------------------------------------------------------------------------------------------------------------
# ABI rules
# ------------------------------------------------------------------------------------------------------------------
def test_calls_of_one_context_on_two_streams_are_ordered(renderer, oracle_lib):
    """generate_hiz on stream A, then cull_geometry(use_hiz) on stream B with no host or event synchronisation in between: the
    context orders its own calls on the device (include/oxcull.h, "Conventions")."""
    spec = SceneSpec(n_mesh_instances=400, meshlets_per_mesh=250, seed=41, with_geometry=False)
    cpu = make_scene(spec, "cpu")
    gpu = cpu.to("cuda")
    depth_cpu = make_depth(2048, 2048, 48, seed=41)
    hz_cpu, levels, offs = oracle_hiz(depth_cpu, 1024, 1024)
    hizd = {"data": hz_cpu, "w": 1024, "h": 1024, "levels": levels, "offs": offs}
    mask0 = torch.zeros((cpu.n_meshlet_instances + 31) // 32, dtype=torch.int32)
    want = oracle_frame(cpu, use_hiz=True, hiz=hizd, mask=mask0, two_pass=True, with_triangles=False)
    depth = ImageAttachment.depth(depth_cpu.cuda())
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for rep in range(6):
        hiz = ImageAttachment.hiz(1024, 1024, "cuda")  # zeros: a cull that overtook the build would see an empty pyramid
        frame = PreparedFrame.create(gpu, with_triangles=False)
        renderer.prepared_frame = frame
        ctx = CullGeometryContext(use_hiz=True, init_cull_meshes=False, cull_flags=L.CULL_TEST_ALL, cull_camera=gpu.cull_camera(), hiz_attachment=hiz, stages=L.STAGE_MESHLETS)
        torch.cuda.synchronize()
        renderer.seed_meshlet_instances(ctx, gpu.n_meshlet_instances, stream=sa)
        renderer.generate_hiz(MainGeometryContext(depth, hiz), stream=sa)
        renderer.cull_geometry(ctx, stream=sb)
        ctx.cull_flags = L.CULL_TEST_ALL | L.CULL_LATE_PASS
        renderer.cull_geometry(ctx, stream=sa)
        c = renderer.read_counters(ctx, stream=sb)
        late = frame.visible_meshlet_instances_indices_buffer[c.early_visible_meshlet_instances:c.early_visible_meshlet_instances + c.cull_triangles_cmd_x].cpu().numpy()
        assert c.early_visible_meshlet_instances == want["early"] and np.array_equal(late, want["late_visible"]), f"rep {rep}"
        assert np.array_equal(frame.meshlet_instance_visibility_mask_buffer.cpu().numpy(), want["mask"])


def test_scratch_growth_is_refused_inside_a_stream_capture():
    """A call that would have to grow the context's scratch returns OXC_INVALID_ARG while its stream is being captured (and
    succeeds, captured, once oxc_reserve has run)."""
    r = RendererInstance(0)  # a fresh context: nothing reserved
    gpu = make_scene(SceneSpec(n_mesh_instances=64, meshlets_per_mesh=128, seed=43, with_geometry=False), "cuda")
    frame = PreparedFrame.create(gpu, with_triangles=False)
    r.prepared_frame = frame
    ctx = CullGeometryContext(init_cull_meshes=True, cull_flags=L.CULL_TEST_ALL, cull_camera=gpu.cull_camera(), stages=L.STAGE_MESHES | L.STAGE_MESHLETS)
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with pytest.raises(L.OxcError) as ei:
        with torch.cuda.graph(g, stream=s):
            r.cull_geometry(ctx, stream=s)
    assert ei.value.status == L.OXC_INVALID_ARG and "captured" in str(ei.value)
    r.reserve(gpu.n_mesh_instances, gpu.n_meshlet_instances)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=s):
        r.cull_geometry(ctx, stream=s)
    g2.replay()
    torch.cuda.synchronize()
    cpu = gpu.to("cpu")
    want = oracle_frame(cpu, run_cull_meshes=True, with_triangles=False)
    c = r.read_counters(ctx)
    assert c.cull_triangles_cmd_x == len(want["visible"])
    assert np.array_equal(frame.visible_meshlet_instances_indices_buffer[:c.cull_triangles_cmd_x].cpu().numpy(), want["visible"])
    r.close()


def test_device_side_list_length_is_clamped_to_the_callers_buffers(renderer, oracle_lib):
    """visibility.total (device memory, e.g. produced by an earlier cull_meshes) larger than frame.max_meshlet_instance_count:
    the kernels stop at the buffers' capacity instead of running past them."""
    spec = SceneSpec(n_mesh_instances=50, meshlets_per_mesh=100, seed=47, with_geometry=False)
    cpu = make_scene(spec, "cpu")
    gpu = cpu.to("cuda")
    N = gpu.n_meshlet_instances
    want = oracle.cull_meshlets(cpu, cpu.cull_camera(), cpu.meshlet_instances)
    frame = PreparedFrame.create(gpu, with_triangles=False)
    guard = torch.full((4096,), 0x5A5A5A5A, dtype=torch.int32, device="cuda")  # (allocated right after the frame's buffers)
    renderer.prepared_frame = frame
    ctx = CullGeometryContext(init_cull_meshes=False, cull_flags=L.CULL_TEST_ALL, cull_camera=gpu.cull_camera(), stages=L.STAGE_MESHLETS)
    renderer.seed_meshlet_instances(ctx, N + 100_000)  # lies about the list length
    renderer.cull_geometry(ctx)
    c = renderer.read_counters(ctx)
    assert c.cull_triangles_cmd_x == want.numel()
    assert torch.equal(frame.visible_meshlet_instances_indices_buffer[:c.cull_triangles_cmd_x].cpu(), want)
    assert bool((guard == 0x5A5A5A5A).all())


def test_visibility_mask_buffer_size_is_validated(renderer):
    gpu = make_scene(SceneSpec(n_mesh_instances=8, meshlets_per_mesh=100, seed=49, with_geometry=False), "cuda")
    frame = PreparedFrame.create(gpu, with_triangles=False)
    frame.meshlet_instance_visibility_mask_buffer = torch.zeros(4, dtype=torch.int32, device="cuda")  # 128 bits for 800 meshlets
    renderer.prepared_frame = frame
    hiz = ImageAttachment.hiz(64, 64, "cuda")
    ctx = CullGeometryContext(use_hiz=True, init_cull_meshes=False, cull_flags=L.CULL_TEST_ALL, cull_camera=gpu.cull_camera(), hiz_attachment=hiz, stages=L.STAGE_MESHLETS)
    renderer.seed_meshlet_instances(ctx, gpu.n_meshlet_instances)
    with pytest.raises(L.OxcError) as ei:
        renderer.cull_geometry(ctx)
    assert ei.value.status == L.OXC_INVALID_ARG


def test_pack_counters_equals_read_counters(renderer, oracle_lib):
    gpu = make_scene(SceneSpec(n_mesh_instances=30, meshlets_per_mesh=90, seed=51), "cuda")
    frame = PreparedFrame.create(gpu)
    renderer.prepared_frame = frame
    ctx = CullGeometryContext(init_cull_meshes=False, cull_flags=L.CULL_TEST_ALL, cull_camera=gpu.cull_camera())
    renderer.seed_meshlet_instances(ctx, gpu.n_meshlet_instances)
    renderer.cull_geometry(ctx)
    packed = torch.zeros(4, dtype=torch.int32, device="cuda")
    renderer.pack_counters(ctx, packed)
    c = renderer.read_counters(ctx)
    assert packed.cpu().tolist() == [c.cull_triangles_cmd_x, c.early_visible_meshlet_instances, c.late_visible_meshlet_instances, c.draw_index_count]
    assert c.cull_triangles_cmd_x > 0 and c.draw_index_count > 0


def test_pack_counters_of_a_context_without_counter_buffers_is_refused(renderer, oracle_lib):
    """oxc_pack_counters / oxc_pack_counters_batch launch a kernel that reads the context's counter buffers: a context no cull_geometry call has
    filled in yet (null buffers) is an argument error, not a fault on the device."""
    gpu = make_scene(SceneSpec(n_mesh_instances=4, meshlets_per_mesh=8, seed=3), "cuda")
    renderer.prepared_frame = PreparedFrame.create(gpu)
    ctx = CullGeometryContext(init_cull_meshes=True, cull_flags=L.CULL_TEST_FRUSTUM, cull_camera=gpu.cull_camera())
    packed = torch.zeros(4, dtype=torch.int32, device="cuda")
    ctx.c()  # (a well-formed struct: what is refused is the missing buffers)
    with pytest.raises(L.OxcError) as ei:
        renderer.pack_counters(ctx, packed)
    assert ei.value.status == L.OXC_INVALID_ARG and "counter buffers" in str(ei.value)
    renderer.cull_geometry(ctx)
    renderer.pack_counters(ctx, packed)  # (filled in now)
