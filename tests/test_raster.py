"""Consumer of the indirect draw (oxc_draw_visbuffer): hand-checkable coverage on the CPU checker,
GPU parity of the packed depth|vis image, and the closed two-pass loop early cull -> draw -> depth -> HiZ -> late cull -> draw."""
import numpy as np
import pytest
import torch

from oxylus_amd import lib as L
from oxylus_amd.synth import SceneSpec, build_meshlets_simple, make_scene, make_scene_from_mesh


def _flat_scene(tris_xy, W, H, z=0.5):
    """One mesh instance, identity world matrix, vertices at the given PIXEL coordinates (half-exact values) and a
    projection_view that maps pixels to NDC with w = 1: screen = (ndc * 0.5 + 0.5) * extent = the pixel coordinate."""
    import oracle

    verts = sorted({tuple(v) for t in tris_xy for v in t})
    index = {v: i for i, v in enumerate(verts)}
    pos = torch.tensor([[x, y, z] for x, y in verts], dtype=torch.float32)
    tris = torch.tensor([[index[tuple(v)] for v in t] for t in tris_xy], dtype=torch.int64)
    meshlets, vidx, micro = build_meshlets_simple(tris)
    b, m6, q = oracle.build_meshlet_bounds(pos, meshlets, vidx, micro)
    s = make_scene_from_mesh(1, b, meshlets, micro, vidx, q, m6, device="cpu")
    s.transforms[0] = torch.eye(4).flatten()
    pv = torch.zeros(4, 4)  # [col][row]
    pv[0, 0], pv[3, 0] = 2.0 / W, -1.0
    pv[1, 1], pv[3, 1] = 2.0 / H, -1.0
    pv[2, 2], pv[3, 3] = 1.0, 1.0
    n_tris = tris.shape[0]
    # index list as cull_triangles writes it: (meshlet instance << 8) | corner, all triangles of meshlet 0
    idx = torch.tensor([(0 << 8) | c for c in range(3 * n_tris)], dtype=torch.int32)
    return s, pv.flatten().tolist(), idx


def _coverage(tris_xy, W=16, H=16):
    import oracle

    s, pv, idx = _flat_scene(tris_xy, W, H)
    vd = torch.zeros((H, W), dtype=torch.int64)
    oracle.draw_visbuffer(s, s.meshlet_instances, idx, pv, W, H, vd)
    depth, vis = oracle.resolve_visbuffer(vd)
    return depth.numpy(), vis.numpy()


def test_front_face_right_triangle_coverage_and_top_left_rule(oracle_lib):
    # screen-space (y down) triangle (2,2) (2,10) (10,2): negative fixed-point area = front facing.
    # Pixel centres (x+.5, y+.5) inside x > 2, y > 2, x + y < 12; the diagonal passes through the centres with
    # x + y = 11 (e.g. pixel (5,5)); after orientation it is a "down or left-going" edge or not -- the stated rule
    # gives it to exactly one of the two triangles that share it (next test); here it is simply counted.
    depth, vis = _coverage([[(2, 2), (2, 10), (10, 2)]])
    inside = {(x, y) for y in range(16) for x in range(16) if x >= 2 and y >= 2 and (x + 0.5) + (y + 0.5) < 12}
    on_diag = {(x, y) for y in range(16) for x in range(16) if x >= 2 and y >= 2 and (x + 0.5) + (y + 0.5) == 12}
    got = {(x, y) for y in range(16) for x in range(16) if depth[y, x] > 0}
    assert inside <= got <= inside | on_diag
    assert np.all(depth[depth > 0] == 0.5) and np.all(vis[depth > 0] == 0)  # (instance 0 << 8) | triangle 0


def test_back_face_is_dropped(oracle_lib):
    depth, _ = _coverage([[(2, 2), (10, 2), (2, 10)]])  # the other winding
    assert not (depth > 0).any()


def test_shared_edge_is_covered_exactly_once(oracle_lib):
    # a quad split along its diagonal: every pixel centre of the quad belongs to exactly one triangle
    import oracle

    a, b = [(2, 2), (2, 10), (10, 2)], [(10, 2), (2, 10), (10, 10)]
    da, _ = _coverage([a])
    db, _ = _coverage([b])
    quad = np.zeros((16, 16), dtype=bool)
    quad[2:10, 2:10] = True
    assert np.array_equal((da > 0) ^ (db > 0), quad) and not ((da > 0) & (db > 0)).any()
    both, vis = _coverage([a, b])
    assert np.array_equal(both > 0, quad)
    assert set(np.unique(vis[quad]).tolist()) == {0, 1}  # triangle ids 0 and 1 of meshlet instance 0


def test_nearer_triangle_wins_and_ties_go_to_the_larger_id(oracle_lib):
    import oracle

    W = H = 16
    s, pv, idx = _flat_scene([[(2, 2), (2, 10), (10, 2)], [(3, 3), (3, 9), (9, 3)]], W, H)
    vd = torch.zeros((H, W), dtype=torch.int64)
    oracle.draw_visbuffer(s, s.meshlet_instances, idx, pv, W, H, vd)
    _, vis = oracle.resolve_visbuffer(vd)
    assert vis[4, 4].item() == 1 and vis[2, 8].item() == 0  # equal depth 0.5 where both cover: the larger vis id wins


def _persp_scene(tris_xyz):
    """Vertices (x, y, w) in a space where projection_view gives clip = (x, y, 0.25, w): ndc = (x/w, y/w), depth = 0.25/w."""
    import oracle

    verts = sorted({tuple(v) for t in tris_xyz for v in t})
    index = {v: i for i, v in enumerate(verts)}
    pos = torch.tensor(verts, dtype=torch.float32)
    tris = torch.tensor([[index[tuple(v)] for v in t] for t in tris_xyz], dtype=torch.int64)
    meshlets, vidx, micro = build_meshlets_simple(tris)
    b, m6, q = oracle.build_meshlet_bounds(pos, meshlets, vidx, micro)
    s = make_scene_from_mesh(1, b, meshlets, micro, vidx, q, m6, device="cpu")
    s.transforms[0] = torch.eye(4).flatten()
    pv = torch.zeros(4, 4)  # [col][row]
    pv[0, 0] = pv[1, 1] = 1.0
    pv[3, 2] = 0.25
    pv[2, 3] = 1.0
    idx = torch.tensor([(0 << 8) | c for c in range(3 * tris.shape[0])], dtype=torch.int32)
    return s, pv.flatten().tolist(), idx


def _ray_coverage(tri, W, H):
    """Independent statement of what a clipped triangle covers: pixel centre -> ray (nx, ny, 1) w through the (x, y, w) space ->
    intersection with the triangle's plane in double precision.  Returns (inside, margin): margin = smallest barycentric."""
    a, b, c = (np.asarray(v, dtype=np.float64) for v in tri)
    n = np.cross(b - a, c - a)
    ys, xs = np.mgrid[0:H, 0:W]
    d = np.stack([((xs + 0.5) / W - 0.5) * 2, ((ys + 0.5) / H - 0.5) * 2, np.ones((H, W))], axis=-1)
    den = d @ n
    with np.errstate(divide="ignore", invalid="ignore"):
        t = (a @ n) / den
        p = d * t[..., None]  # homogeneous point on the plane; its w is t
        area = np.dot(n, n)
        l0 = np.cross(c - b, p - b) @ n / area
        l1 = np.cross(a - c, p - c) @ n / area
        l2 = np.cross(b - a, p - a) @ n / area
    margin = np.minimum(np.minimum(l0, l1), l2)
    ok = np.isfinite(t) & (t >= 2.0 ** -10)
    return ok, np.where(ok, margin, -1.0), np.where(ok, 0.25 / np.where(ok, t, 1.0), 0.0)


@pytest.mark.parametrize("tri", [
    [(-0.5, -0.5, 1.0), (0.0, 1.0, -0.5), (0.5, -0.5, 1.0)],      # one corner behind the camera: round 1 dropped it whole
    [(-0.25, -0.75, 2.0), (-0.5, 0.25, -0.5), (0.5, 0.0, -0.25)],  # two corners behind
    [(-0.5, -0.5, 1.0), (0.0, 300.0, 1.0), (0.5, -0.5, 1.0)],      # crosses the guard band (|y| > 64 w), all w > 0
    [(-0.5, -0.5, 1.0), (0.0, 0.5, 0.0), (0.5, -0.5, 1.0)],        # a corner exactly on the camera plane
], ids=["one-behind", "two-behind", "guard-band", "w-zero"])
def test_triangle_crossing_the_camera_plane_is_clipped_not_dropped(oracle_lib, tri):
    import oracle

    W, H = 96, 64
    for winding in (tri, [tri[0], tri[2], tri[1]]):
        s, pv, idx = _persp_scene([winding])
        vd = torch.zeros((H, W), dtype=torch.int64)
        oracle.draw_visbuffer(s, s.meshlet_instances, idx, pv, W, H, vd)
        depth, _ = oracle.resolve_visbuffer(vd)
        depth = depth.numpy()
        ok, margin, want_depth = _ray_coverage(winding, W, H)
        if not (depth > 0).any():
            continue  # this winding faces away
        # pixels clearly inside must be covered, pixels clearly outside must not; a band of 1e-3 around the edges (snapping to 1/256 px
        # and the 2^-10 camera-plane offset) is left to the exact GPU parity tests
        want_in = (margin > 2e-3) & (want_depth < 0.999)  # depth > 1 (nearer than w = 0.25 here) is rejected by the depth range rule
        assert (depth > 0)[want_in].all()
        assert not (depth > 0)[ok & (margin < -2e-3)].any() and not (depth > 0)[~ok].any()
        sel = want_in & (want_depth <= 1.0)
        assert np.allclose(depth[sel], want_depth[sel], rtol=2e-3)
        break
    else:
        pytest.fail("neither winding produced coverage")


def test_clipped_neighbours_share_their_new_vertices(oracle_lib):
    """Two triangles sharing an edge that crosses the camera plane: every pixel of the clipped quad belongs to exactly one of them."""
    quad = [(-0.5, -0.5, 1.0), (0.5, -0.5, 1.0), (0.4, 1.0, -0.5), (-0.6, 1.0, -0.5)]
    a, b = [quad[0], quad[2], quad[1]], [quad[0], quad[3], quad[2]]
    import oracle

    W, H = 128, 128

    def cover(tris):
        s, pv, idx = _persp_scene(tris)
        vd = torch.zeros((H, W), dtype=torch.int64)
        oracle.draw_visbuffer(s, s.meshlet_instances, idx, pv, W, H, vd)
        return oracle.resolve_visbuffer(vd)[0].numpy() > 0

    ca, cb = cover([a]), cover([b])
    if not ca.any():
        a, b = [a[0], a[2], a[1]], [b[0], b[2], b[1]]
        ca, cb = cover([a]), cover([b])
    assert ca.any() and cb.any() and not (ca & cb).any()
    both = cover([a, b])
    assert np.array_equal(both, ca | cb)
    # no crack along the shared edge: every row's covered span is contiguous
    for y in range(H):
        xs = np.flatnonzero(both[y])
        if xs.size:
            assert xs[-1] - xs[0] + 1 == xs.size, y


def _gpu_vs_oracle(renderer, cpu, gpu, W, H, wide=False, max_tris=64):
    import oracle
    from oxylus_amd.renderer import CullGeometryContext, PreparedFrame

    cam = cpu.cull_camera()
    pv = [cam.projection_view[i] for i in range(16)]
    want_vis = oracle.cull_meshlets(cpu, cam, cpu.meshlet_instances)
    want_idx = oracle.cull_triangles(cpu, cam, cpu.meshlet_instances, want_vis, 0, want_vis.numel(), wide=wide)
    vd = torch.zeros((H, W), dtype=torch.int64)
    oracle.draw_visbuffer(cpu, cpu.meshlet_instances, want_idx, pv, W, H, vd, wide=wide)
    frame = PreparedFrame.create(gpu, max_tris=max_tris, index_words=2 if int(wide) == 2 else 1)
    renderer.prepared_frame = frame
    ctx = CullGeometryContext(init_cull_meshes=False, cull_flags=L.CULL_TEST_ALL, cull_camera=gpu.cull_camera(), wide_triangle_index=int(wide))
    renderer.seed_meshlet_instances(ctx, gpu.n_meshlet_instances)
    renderer.cull_geometry(ctx)
    got = torch.full((H, W), -1, dtype=torch.int64, device="cuda")
    vis = torch.empty((H, W), dtype=torch.int32, device="cuda")
    renderer.draw_visbuffer(ctx, pv, W, H, got, clear=True, visbuffer=vis)
    torch.cuda.synchronize()
    assert torch.equal(got.cpu(), vd)
    assert torch.equal(vis.cpu(), (vd & 0xFFFFFFFF).to(torch.int32))
    return vd


@pytest.mark.gpu
@pytest.mark.parametrize("spec,W,H", [(SceneSpec(n_mesh_instances=30, meshlets_per_mesh=40, seed=5, scene_depth=40.0), 256, 256),
                                      (SceneSpec(n_mesh_instances=8, meshlets_per_mesh=20, seed=6, scene_depth=8.0), 333, 127),      # close-up: large triangles
                                      (SceneSpec(n_mesh_instances=12, meshlets_per_mesh=30, seed=7, scene_depth=30.0, ragged=True), 64, 64)],
                         ids=["256x256", "closeup-333x127", "ragged-64x64"])
def test_gpu_draw_matches_oracle(renderer, oracle_lib, spec, W, H):
    cpu = make_scene(spec, "cpu")
    vd = _gpu_vs_oracle(renderer, cpu, cpu.to("cuda"), W, H)
    assert (vd != 0).any()


def _draw_list_both(renderer, cpu, idx, pv, W, H):
    """Draws a hand-written triangle list (what cull_triangles would have put into reordered_indices_buffer) on both sides."""
    import oracle
    from oxylus_amd.renderer import CullGeometryContext, PreparedFrame

    gpu = cpu.to("cuda")
    frame = PreparedFrame.create(gpu, max_tris=64)
    renderer.prepared_frame = frame
    assert frame.reordered_indices_buffer.numel() >= idx.numel()
    frame.reordered_indices_buffer[:idx.numel()] = idx.cuda()
    cmd = torch.tensor([idx.numel(), 1, 0, 0, 0], dtype=torch.int32, device="cuda")
    ctx = CullGeometryContext(init_cull_meshes=False, cull_flags=0, cull_camera=gpu.cull_camera())
    got = torch.empty((H, W), dtype=torch.int64, device="cuda")
    renderer.draw_visbuffer(ctx, pv, W, H, got, clear=True, draw_cmd=cmd)
    torch.cuda.synchronize()
    vd = torch.zeros((H, W), dtype=torch.int64)
    oracle.draw_visbuffer(cpu, cpu.meshlet_instances, idx, pv, W, H, vd)
    return got.cpu(), vd


@pytest.mark.gpu
@pytest.mark.parametrize("depth,W,H", [(2.0, 640, 360), (0.5, 1920, 1080)], ids=["640x360", "1920x1080"])
def test_gpu_draw_with_the_camera_inside_the_geometry(renderer, oracle_lib, depth, W, H):
    """Every triangle of every meshlet instance (no culling in front of the draw), instances all around the camera: hundreds of
    triangles cross the camera plane and the guard band."""
    import oracle

    cpu = make_scene(SceneSpec(n_mesh_instances=24, meshlets_per_mesh=20, seed=9, scene_depth=depth), "cpu")
    cam = cpu.cull_camera()
    pv = [cam.projection_view[i] for i in range(16)]
    tris = cpu.meshlets[cpu.meshlet_instances[:, 1].long(), 3].tolist()
    idx = torch.tensor([(i << 8) | c for i, t in enumerate(tris) for c in range(3 * t)], dtype=torch.int32)
    got, vd = _draw_list_both(renderer, cpu, idx, pv, W, H)
    assert oracle.draw_clipped_count() > 500, oracle.draw_clipped_count()
    assert torch.equal(got, vd) and (vd != 0).any()
    st = renderer.debug_raster_stats()
    assert st["clipped"] > 500 and st["big"] > 0, st


@pytest.mark.gpu
def test_gpu_draw_when_the_big_list_overflows(oracle_lib):
    """A context whose big list holds 8192 triangles (oxc_debug_set_tuning before the first draw) and a scene with far more
    triangles whose pixel boxes exceed 8 x 8: segments fill up and the setup lanes walk the excess themselves -- slow, and still
    the same image."""
    import oracle
    from oxylus_amd.renderer import RendererInstance

    from oxylus_amd import lib as L

    r = RendererInstance(0)
    r.debug_set_tuning(L.TUNE_RASTER_BIG_CAPACITY, 8192)
    try:
        cpu = make_scene(SceneSpec(n_mesh_instances=60, meshlets_per_mesh=50, seed=12, scene_depth=12.0), "cpu")
        cam = cpu.cull_camera()
        pv = [cam.projection_view[i] for i in range(16)]
        W, H = 1920, 1080
        want_vis = oracle.cull_meshlets(cpu, cam, cpu.meshlet_instances)  # frustum + cone: nothing near or behind the camera plane
        idx = oracle.cull_triangles(cpu, cam, cpu.meshlet_instances, want_vis, 0, want_vis.numel())
        got, vd = _draw_list_both(r, cpu, idx, pv, W, H)
        st = r.debug_raster_stats()
        assert st["big"] > 8192 and st["overflowed_segments"] > 0, st
        assert torch.equal(got, vd) and (vd != 0).any()
    finally:
        r.close()


@pytest.mark.gpu
def test_gpu_draw_when_the_clip_queue_overflows(oracle_lib):
    """The id queue of the triangles that cross a clip plane holds 4096 entries here; the camera sits inside a scene in which more
    than that cross: the overflow pass walks the index list again and clips what the queue could not record -- same image."""
    import oracle
    from oxylus_amd.renderer import RendererInstance

    from oxylus_amd import lib as L

    r = RendererInstance(0)
    r.debug_set_tuning(L.TUNE_RASTER_BIG_CAPACITY, 4096)
    try:
        cpu = make_scene(SceneSpec(n_mesh_instances=160, meshlets_per_mesh=40, seed=19, scene_depth=0.5), "cpu")
        cam = cpu.cull_camera()
        pv = [cam.projection_view[i] for i in range(16)]
        tris = cpu.meshlets[cpu.meshlet_instances[:, 1].long(), 3].tolist()
        idx = torch.tensor([(i << 8) | c for i, t in enumerate(tris) for c in range(3 * t)], dtype=torch.int32)
        got, vd = _draw_list_both(r, cpu, idx, pv, 480, 270)
        assert oracle.draw_clipped_count() > 4096, oracle.draw_clipped_count()
        st = r.debug_raster_stats()
        assert st["clipped"] > 4096, st
        assert torch.equal(got, vd) and (vd != 0).any()
    finally:
        r.close()


@pytest.mark.gpu
def test_gpu_clipped_triangles_match_oracle(renderer, oracle_lib):
    """The hand-made clip cases of the CPU tests (one / two corners behind the camera, guard band, w = 0) drawn by the device."""
    import oracle

    tris = [[(-0.5, -0.5, 1.0), (0.5, -0.5, 1.0), (0.0, 1.0, -0.5)], [(-0.25, -0.75, 2.0), (0.5, 0.0, -0.25), (-0.5, 0.25, -0.5)],
            [(-0.5, -0.5, 1.0), (0.5, -0.5, 1.0), (0.0, 300.0, 1.0)], [(-0.5, -0.5, 1.5), (0.5, -0.5, 1.5), (0.0, 0.5, 0.0)]]
    tris = tris + [[t[0], t[2], t[1]] for t in tris]  # both windings
    W, H = 200, 120
    s, pv, idx = _persp_scene(tris)
    vd = torch.zeros((H, W), dtype=torch.int64)
    oracle.draw_visbuffer(s, s.meshlet_instances, idx, pv, W, H, vd)
    assert oracle.draw_clipped_count() >= 4 and (vd != 0).any()
    got, vd2 = _draw_list_both(renderer, s, idx, pv, W, H)
    assert torch.equal(vd2, vd)
    assert torch.equal(got, vd)


@pytest.mark.gpu
def test_gpu_draw_wide_index(renderer, oracle_lib):
    spec = SceneSpec(n_mesh_instances=10, meshlets_per_mesh=20, tris_per_meshlet=124, seed=8, scene_depth=30.0)
    cpu = make_scene(spec, "cpu")
    vd = _gpu_vs_oracle(renderer, cpu, cpu.to("cuda"), 200, 200, wide=True, max_tris=128)
    assert (vd != 0).any()
    # SURVEY A.7's pair form (wide_triangle_index = 2): instance = pair.x, corner = pair.y instead of the shifts of visbuffer.slang:9-14 -- the same
    # triangles, so the same image (vis = (instance << 8) | corner / 3 as VisBufferData::encode either way)
    vd2 = _gpu_vs_oracle(renderer, cpu, cpu.to("cuda"), 200, 200, wide=2, max_tris=128)
    assert torch.equal(vd2, vd)


@pytest.mark.gpu
def test_two_pass_loop_early_draw_hiz_late_draw(renderer, oracle_lib):
    """The sequence of RendererInstance::render (RendererInstance.cpp:842-884) without a graphics queue: early
    cull with last frame's mask -> draw -> depth -> generate_hiz -> late cull -> draw on top.  Every intermediate
    (lists, mask, HiZ bytes, final depth|vis image) must equal the checker's."""
    import oracle
    from oxylus_amd.renderer import CullGeometryContext, ImageAttachment, MainGeometryContext, PreparedFrame
    from oxylus_amd.synth import hiz_layout
    from util import oracle_hiz

    W = H = 512
    spec = SceneSpec(n_mesh_instances=60, meshlets_per_mesh=50, seed=12, scene_depth=50.0)
    cpu = make_scene(spec, "cpu")
    gpu = cpu.to("cuda")
    cam = cpu.cull_camera()
    pv = [cam.projection_view[i] for i in range(16)]
    N = cpu.n_meshlet_instances
    g = torch.Generator().manual_seed(3)
    bits = (torch.rand(((N + 31) // 32, 32), generator=g) < 0.5).to(torch.int64)
    mask0 = (bits << torch.arange(32)).sum(1).to(torch.int32)

    # ---- checker chain
    mask = mask0.clone()
    zero_hiz_data, levels, offs = oracle_hiz(torch.zeros((H, W)), W // 2, H // 2)
    hz0 = oracle.make_hiz(zero_hiz_data.numpy(), W // 2, H // 2, levels, offs)
    vis_cnt = oracle.Visibility(N, 0, 0)
    out = torch.zeros(N, dtype=torch.int32)
    n_e = oracle.cull_meshlets_hiz(cpu, cam, cpu.meshlet_instances, L.CULL_TEST_ALL, hz0, vis_cnt, mask, out)
    e_vis = out[:n_e].clone()
    e_idx = oracle.cull_triangles(cpu, cam, cpu.meshlet_instances, out, 0, n_e)
    vd = torch.zeros((H, W), dtype=torch.int64)
    oracle.draw_visbuffer(cpu, cpu.meshlet_instances, e_idx, pv, W, H, vd)
    depth, _ = oracle.resolve_visbuffer(vd)
    hiz_data, levels, offs = oracle_hiz(depth, W // 2, H // 2)
    hz = oracle.make_hiz(hiz_data.numpy(), W // 2, H // 2, levels, offs)
    n_l = oracle.cull_meshlets_hiz(cpu, cam, cpu.meshlet_instances, L.CULL_TEST_ALL | L.CULL_LATE_PASS, hz, vis_cnt, mask, out)
    l_vis = out[vis_cnt.early: vis_cnt.early + n_l].clone()
    l_idx = oracle.cull_triangles(cpu, cam, cpu.meshlet_instances, out, vis_cnt.early, n_l)
    oracle.draw_visbuffer(cpu, cpu.meshlet_instances, l_idx, pv, W, H, vd)

    # ---- HIP chain
    frame = PreparedFrame.create(gpu)
    frame.meshlet_instance_visibility_mask_buffer.copy_(mask0.cuda())
    renderer.prepared_frame = frame
    hiz_att = ImageAttachment.hiz(W // 2, H // 2, "cuda")   # all zero: the previous frame's pyramid of an empty frame
    ctx = CullGeometryContext(use_hiz=True, init_cull_meshes=False, cull_flags=L.CULL_TEST_ALL, cull_camera=gpu.cull_camera(), hiz_attachment=hiz_att)
    renderer.seed_meshlet_instances(ctx, N)
    renderer.cull_geometry(ctx)
    c = renderer.read_counters(ctx)
    assert torch.equal(frame.visible_meshlet_instances_indices_buffer[: c.cull_triangles_cmd_x].cpu(), e_vis)
    got = torch.empty((H, W), dtype=torch.int64, device="cuda")
    depth_att = ImageAttachment.depth(torch.zeros((H, W), dtype=torch.float32, device="cuda"))
    renderer.draw_visbuffer(ctx, pv, W, H, got, clear=True, depth=depth_att)
    renderer.generate_hiz(MainGeometryContext(depth_attachment=depth_att, hiz_attachment=hiz_att))
    assert torch.equal(hiz_att.data.cpu(), hiz_data)
    ctx.cull_flags = L.CULL_TEST_ALL | L.CULL_LATE_PASS
    renderer.cull_geometry(ctx)
    c = renderer.read_counters(ctx)
    first = c.early_visible_meshlet_instances
    assert torch.equal(frame.visible_meshlet_instances_indices_buffer[first: first + c.cull_triangles_cmd_x].cpu(), l_vis)
    renderer.draw_visbuffer(ctx, pv, W, H, got, clear=False)
    torch.cuda.synchronize()
    assert torch.equal(frame.meshlet_instance_visibility_mask_buffer.cpu(), mask)
    assert torch.equal(got.cpu(), vd)
    assert e_vis.numel() > 0 and l_vis.numel() > 0 and (vd != 0).float().mean() > 0.01
