"""GPU parity: the HIP path (through the C ABI) vs the CPU oracle on the same seeded scenes.

Bit-exact bar: counts equal, visibility-mask bytes equal, HiZ pyramid bytes equal, and -- because
both sides emit in ascending order -- visible-meshlet and packed-triangle index arrays
byte-identical WITHOUT sorting.
"""
import numpy as np
import pytest
import torch

from oxylus_amd import lib as L
from oxylus_amd.renderer import ImageAttachment, MainGeometryContext
from oxylus_amd.synth import SceneSpec, make_depth, make_scene

from util import assert_same, gpu_frame, oracle_frame, oracle_hiz

pytestmark = pytest.mark.gpu


def _pair(spec):
    cpu = make_scene(spec, "cpu")
    return cpu, cpu.to("cuda")


@pytest.mark.parametrize(
    "spec",
    [
        SceneSpec(n_mesh_instances=64, meshlets_per_mesh=256),
        SceneSpec(n_mesh_instances=27, meshlets_per_mesh=100, seed=7),          # waves straddle instances
        SceneSpec(n_mesh_instances=200, meshlets_per_mesh=3, seed=9),           # many instances per wave
        SceneSpec(n_mesh_instances=8, meshlets_per_mesh=1000, seed=11, nonuniform_scale=True),
        SceneSpec(n_mesh_instances=5, meshlets_per_mesh=513, seed=13, ragged=True),
        SceneSpec(n_mesh_instances=1, meshlets_per_mesh=1, seed=15),
        SceneSpec(n_mesh_instances=64, meshlets_per_mesh=64, seed=17, share_meshes=4),
    ],
    ids=["64x256", "27x100", "200x3", "8x1000-nonuniform", "5x513-ragged", "1x1", "instanced"],
)
def test_cull_meshlets_and_triangles(renderer, oracle_lib, spec):
    cpu, gpu = _pair(spec)
    want = oracle_frame(cpu)
    got = gpu_frame(renderer, gpu)
    assert_same(want, got, ["total", "visible", "indices"])
    assert len(want["visible"]) > 0 or spec.n_mesh_instances == 1


def test_cull_meshes_lod_select_and_expansion(renderer, oracle_lib):
    spec = SceneSpec(n_mesh_instances=300, meshlets_per_mesh=96, lod_count=4, seed=21)
    cpu, gpu = _pair(spec)
    want = oracle_frame(cpu, run_cull_meshes=True)
    got = gpu_frame(renderer, gpu, run_cull_meshes=True)
    assert_same(want, got, ["total", "cull_meshlets_cmd_x", "lod_index", "meshlet_instances", "visible", "indices"])
    assert len(set(want["lod_index"].tolist())) > 1, "scene should exercise several LODs"


def test_cull_meshes_frustum_only_flags(renderer, oracle_lib):
    spec = SceneSpec(n_mesh_instances=100, meshlets_per_mesh=40, lod_count=3, seed=23)
    cpu, gpu = _pair(spec)
    want = oracle_frame(cpu, cull_flags=L.CULL_TEST_FRUSTUM, run_cull_meshes=True)
    got = gpu_frame(renderer, gpu, cull_flags=L.CULL_TEST_FRUSTUM, run_cull_meshes=True)
    assert_same(want, got, ["total", "lod_index", "meshlet_instances", "visible", "indices"])
    assert (want["lod_index"] == 0).all()


@pytest.mark.parametrize("size,depth_scale", [(64, 2), (256, 2), (1024, 2), (256, 1), (128, 3), (32, 2), (8, 2), (4096, 2), (2048, 1)],
                         ids=["64", "256", "1024", "256-same-size", "128-x3", "32", "8", "4096-max-13-mips", "2048"])
def test_generate_hiz(renderer, oracle_lib, size, depth_scale):
    depth = make_depth(size * depth_scale, size * depth_scale, 24, seed=size)
    depth += torch.rand(depth.shape, generator=torch.Generator().manual_seed(size)) * 1e-4  # no flat areas
    want, levels, offs = oracle_hiz(depth, size, size)
    if size == 4096:
        assert levels == 13  # hiz.slang binds at most 13 mips (CullGeometry.cpp:24,36-38)
    hiz = ImageAttachment.hiz(size, size, "cuda")
    renderer.generate_hiz(MainGeometryContext(ImageAttachment.depth(depth.cuda()), hiz))
    got = hiz.data.cpu()
    for k in range(levels):
        w = max(1, size >> k)
        a = want[offs[k] // 4: offs[k] // 4 + w * w].numpy().view(np.uint32)
        b = got[offs[k] // 4: offs[k] // 4 + w * w].numpy().view(np.uint32)
        assert np.array_equal(a, b), f"mip {k}: {(a != b).sum()} texels differ"


@pytest.mark.parametrize("w,h,scale", [(512, 256, 2), (192, 320, 2), (64, 1024, 1), (4032, 2112, 1), (100, 36, 2)],
                         ids=["512x256", "192x320-odd-upper-mips", "64x1024", "4032x2112-odd-upper-mips", "100x36-generic"])
def test_generate_hiz_non_square(renderer, oracle_lib, w, h, scale):
    """The tile path (w, h multiples of 64) hands mip 6 to the single-block tail; upper mips of non-power-of-two sizes clamp at the edge."""
    depth = torch.rand((scale * h, scale * w), generator=torch.Generator().manual_seed(5))
    from oxylus_amd.synth import hiz_layout
    import oracle

    levels, offs, total = hiz_layout(w, h)
    want = torch.zeros(total // 4)
    oracle.generate_hiz(depth, want, w, h, levels, offs)
    hiz = ImageAttachment.hiz(w, h, "cuda")
    renderer.generate_hiz(MainGeometryContext(ImageAttachment.depth(depth.cuda()), hiz))
    assert np.array_equal(want.numpy().view(np.uint32), hiz.data.cpu().numpy().view(np.uint32))


def _hiz_pair(size, seed):
    depth = make_depth(2 * size, 2 * size, 48, seed=seed)
    data, levels, offs = oracle_hiz(depth, size, size)
    att = ImageAttachment.hiz(size, size, "cuda")
    att.data.copy_(data.cuda())
    return {"data": data, "w": size, "h": size, "levels": levels, "offs": offs}, att


@pytest.mark.parametrize(
    "spec,size",
    [
        (SceneSpec(n_mesh_instances=64, meshlets_per_mesh=256, seed=31), 256),
        (SceneSpec(n_mesh_instances=27, meshlets_per_mesh=100, seed=33), 512),
        (SceneSpec(n_mesh_instances=150, meshlets_per_mesh=7, seed=35, ragged=True), 128),
    ],
    ids=["64x256", "27x100", "150x7"],
)
def test_two_pass_occlusion_first_frame(renderer, oracle_lib, spec, size):
    """Frame 0: mask all zero => early emits nothing, late emits everything unoccluded."""
    cpu, gpu = _pair(spec)
    hz_cpu, hz_gpu = _hiz_pair(size, spec.seed)
    mask = torch.zeros((cpu.n_meshlet_instances + 31) // 32, dtype=torch.int32)
    want = oracle_frame(cpu, use_hiz=True, hiz=hz_cpu, mask=mask, two_pass=True)
    got = gpu_frame(renderer, gpu, use_hiz=True, hiz=hz_gpu, mask=mask, two_pass=True)
    assert_same(want, got, ["early", "late", "early_visible", "late_visible", "early_indices", "late_indices", "mask"])
    assert want["early"] == 0 and want["late"] > 0


def test_two_pass_occlusion_random_prior_mask(renderer, oracle_lib):
    """Config 3 restatement: given HiZ + given prior-visibility mask -> early + late pass."""
    spec = SceneSpec(n_mesh_instances=40, meshlets_per_mesh=333, seed=41)
    cpu, gpu = _pair(spec)
    hz_cpu, hz_gpu = _hiz_pair(512, 41)
    g = torch.Generator().manual_seed(41)
    words = (cpu.n_meshlet_instances + 31) // 32
    bits = (torch.rand((words, 32), generator=g) < 0.3).to(torch.int64)
    mask = (bits << torch.arange(32)).sum(1).to(torch.int32)
    want = oracle_frame(cpu, use_hiz=True, hiz=hz_cpu, mask=mask, two_pass=True)
    got = gpu_frame(renderer, gpu, use_hiz=True, hiz=hz_gpu, mask=mask, two_pass=True)
    assert_same(want, got, ["early", "late", "early_visible", "late_visible", "early_indices", "late_indices", "mask"])
    assert want["early"] > 0 and want["late"] > 0


def test_two_pass_static_scene_second_frame(renderer, oracle_lib):
    """Frame 1 with frame 0's mask: early U late == frame-0 visible set (static scene)."""
    spec = SceneSpec(n_mesh_instances=32, meshlets_per_mesh=200, seed=43)
    cpu, gpu = _pair(spec)
    hz_cpu, hz_gpu = _hiz_pair(256, 43)
    mask0 = torch.zeros((cpu.n_meshlet_instances + 31) // 32, dtype=torch.int32)
    f0 = gpu_frame(renderer, gpu, use_hiz=True, hiz=hz_gpu, mask=mask0, two_pass=True)
    f1 = gpu_frame(renderer, gpu, use_hiz=True, hiz=hz_gpu, mask=torch.from_numpy(f0["mask"]), two_pass=True)
    assert f1["late"] == 0
    assert np.array_equal(np.sort(f0["late_visible"]), np.sort(f1["early_visible"]))
    assert np.array_equal(f0["mask"], f1["mask"])
    w1 = oracle_frame(cpu, use_hiz=True, hiz=hz_cpu, mask=torch.from_numpy(f0["mask"]), two_pass=True)
    assert_same(w1, f1, ["early", "late", "early_visible", "early_indices", "mask"])


def test_hiz_path_without_occlusion_flag(renderer, oracle_lib):
    spec = SceneSpec(n_mesh_instances=16, meshlets_per_mesh=128, seed=47)
    cpu, gpu = _pair(spec)
    hz_cpu, hz_gpu = _hiz_pair(128, 47)
    mask = torch.full(((cpu.n_meshlet_instances + 31) // 32,), 0x5A5A5A5A, dtype=torch.int32)
    want = oracle_frame(cpu, cull_flags=L.CULL_TEST_FRUSTUM, use_hiz=True, hiz=hz_cpu, mask=mask)
    got = gpu_frame(renderer, gpu, cull_flags=L.CULL_TEST_FRUSTUM, use_hiz=True, hiz=hz_gpu, mask=mask)
    assert_same(want, got, ["early", "early_visible", "early_indices", "mask"])


def test_empty_scene_and_zero_visible(renderer, oracle_lib):
    spec = SceneSpec(n_mesh_instances=4, meshlets_per_mesh=64, seed=51)
    cpu, gpu = _pair(spec)
    # camera looking away: nothing survives the frustum
    for s in (cpu, gpu):
        s.transforms[:, 14] = 5000.0
    want = oracle_frame(cpu)
    got = gpu_frame(renderer, gpu)
    assert len(want["visible"]) == 0
    assert_same(want, got, ["visible", "indices"])


def test_decode_known_answers_all_halfs_and_s8(renderer, oracle_lib):
    """Appendix B.1: every u16 through dequantize_half, every s8 through /127, device vs oracle."""
    import oracle

    h = torch.arange(65536, dtype=torch.int32)
    b = torch.zeros((65536, 8), dtype=torch.int32)
    b[:, 0] = h
    b[:, 1] = (h * 7 + 3) & 0xFFFF
    b[:, 2] = 65535 - h
    b[:, 4] = (h * 13) & 0xFFFF
    b[:, 5] = h ^ 0x8000
    b[:, 6] = (h + 0x7C00) & 0xFFFF
    b[:, 3] = h & 0xFFFF          # cone_axis_xy sweep both bytes
    b[:, 7] = (h * 257) & 0xFFFF  # cone_axis_z / cutoff
    b16 = b.to(torch.int16).contiguous()
    got = renderer.debug_decode_bounds(b16.cuda()).cpu().numpy()
    want = oracle.decode_bounds(b16.numpy())
    # bit-exact everywhere except the payload of NaNs: the device converts with v_cvt_f32_f16, which
    # quiets signalling NaNs (class preserved; no cull decision can depend on a NaN payload)
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(got), nan)
    assert np.array_equal(want.view(np.uint32)[~nan], got.view(np.uint32)[~nan])
    quiet = want.view(np.uint32)[nan] | 0x00400000
    assert np.array_equal(quiet, got.view(np.uint32)[nan] | 0x00400000)


def _vsm_inputs(seed, page_density, other_lights=()):
    from oxylus_amd.renderer import HpbAttachment
    from oxylus_amd.synth import pack_clipmaps, virtual_shadow_matrices

    light = np.array([0.3, -1.0, 0.2])
    light /= np.linalg.norm(light)
    mats, offs, zn = virtual_shadow_matrices([3.0, 1.0, -60.0], light, 500.0, 10.0, 10)
    for v in other_lights:  # views whose frustum normals are NOT clipmap 0's: the kernel's per-view frustum path (round 4)
        l2 = np.array([0.3 + 0.05 * v, -1.0, 0.2 - 0.04 * v])
        m2, o2, _ = virtual_shadow_matrices([3.0, 1.0, -60.0], l2 / np.linalg.norm(l2), 500.0, 10.0, 10)
        mats[v], offs[v] = m2[v], o2[v]
    clip = pack_clipmaps(mats, offs, zn)
    hpb = HpbAttachment.create(64, 64, 10, 7, "cpu")
    g = torch.Generator().manual_seed(seed)
    hpb.level(0).copy_((torch.rand((10, 64, 64), generator=g) < page_density).to(torch.uint8))
    hpb.build_mips()
    return light, mats, clip, zn, hpb


@pytest.mark.parametrize("density,dirty,m,k", [(0.15, [1, 1, 0, 1, 1, 1, 0, 1, 1, 1], 90, 77), (0.02, [1] * 10, 90, 77), (1.0, [0, 0, 0, 0, 1, 0, 0, 0, 0, 0], 90, 77),
                                               (0.5, [0] * 10, 90, 77), (0.1, [1, 0, 1, 1, 1, 1, 1, 0, 1, 1], 700, 600), (0.3, [1] * 10, 3, 5),
                                               (0.15, [1] * 10, 91, 77), (0.15, [0, 1, 1, 1, 0, 1, 1, 1, 1, 1], 92, 77)],
                         ids=["d0.15", "d0.02", "one-view", "nothing-dirty", "420k-meshlets-more-steps-than-counters", "15-meshlets",
                              "three-views-of-other-lights", "every-view-its-own-light"])
def test_cull_meshlets_hpb_multi_view(renderer, oracle_lib, density, dirty, m, k):
    """VSM path (Shadowmaps.cpp:433-463): cull_meshes with TestFrustum against the coarsest
    clipmap, cull_meshlets_hpb over the dirty clipmap views, cull_triangles."""
    import oracle
    from oxylus_amd.renderer import CullGeometryContext, HpbAttachment, PreparedFrame

    spec = SceneSpec(n_mesh_instances=m, meshlets_per_mesh=k, lod_count=2 if m in (90, 91, 92) else 1, seed=61, scene_depth=150.0)
    cpu, gpu = _pair(spec)
    light, mats, clip, zn, hpb = _vsm_inputs(61, density, other_lights={91: (2, 5, 6), 92: tuple(range(1, 10))}.get(m, ()))
    dirty_t = torch.tensor(dirty, dtype=torch.int32)

    def camera(scene):
        cam = scene.cull_camera()
        for i in range(16):
            cam.projection_view[i] = float(mats[9][i])  # coarsest clipmap
        for i in range(3):
            cam.position[i] = float(-light[i])
        cam.near_clip = zn
        return cam

    # oracle
    cam = camera(cpu)
    mli, cmd = oracle.cull_meshes(cpu, cam, L.CULL_TEST_FRUSTUM)
    h = oracle.make_hpb(hpb.data, 64, 64, 10, 7, hpb.level_offset)
    want_vis = oracle.cull_meshlets_hpb(cpu, cam, mli, clip, dirty_t, h)
    want_idx = oracle.cull_triangles(cpu, cam, mli, want_vis, 0, want_vis.numel())
    # HIP
    frame = PreparedFrame.create(gpu, expand=False)
    renderer.prepared_frame = frame
    hpb_gpu = HpbAttachment(hpb.data.cuda(), 64, 64, 10, 7, hpb.level_offset)
    ctx = CullGeometryContext(use_hpb=True, init_cull_meshes=True, cull_flags=L.CULL_TEST_FRUSTUM, cull_camera=camera(gpu), hpb_attachment=hpb_gpu,
                              vsm_clipmaps_buffer=clip.cuda(), vsm_clipmap_dirty_flags_buffer=dirty_t.cuda(), vsm_clipmap_count=10)
    renderer.cull_geometry(ctx)
    c = renderer.read_counters(ctx)
    assert c.total_visible_meshlet_instances == mli.shape[0]
    assert np.array_equal(frame.meshlet_instances_buffer[: mli.shape[0]].cpu().numpy(), mli.numpy())
    got_vis = frame.visible_meshlet_instances_indices_buffer[: c.cull_triangles_cmd_x].cpu()
    assert torch.equal(got_vis, want_vis)
    got_idx = frame.reordered_indices_buffer[: c.draw_index_count].cpu()
    assert torch.equal(got_idx, want_idx)
    if sum(dirty) == 0:
        assert want_vis.numel() == 0
    elif density >= 0.15:
        assert want_vis.numel() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("huge_view", [None, 3], ids=["odd-boxes", "odd-boxes+a-view-scaled-by-2^101"])
def test_cull_meshlets_hpb_non_finite_boxes_and_huge_matrices(renderer, oracle_lib, huge_view):
    """The page test of k_cull_meshlets_hpb_test takes project_aabb's orthographic short cut only while nothing can leave the finite range (a per-lane
    test; round 6 measured hoisting it -- once per candidate + once per view -- and it was no faster: 318.6 against ~309 us).  Boxes whose centre /
    extent hold Inf or NaN halves, and a view whose matrix is scaled beyond 2^100, must come out as the checker's plain arithmetic has them (the general path)."""
    import oracle
    from oxylus_amd.renderer import CullGeometryContext, HpbAttachment, PreparedFrame
    from oxylus_amd.synth import pack_clipmaps

    spec = SceneSpec(n_mesh_instances=60, meshlets_per_mesh=90, seed=67, scene_depth=150.0)
    cpu = make_scene(spec, "cpu")
    g = torch.Generator().manual_seed(7)
    r = torch.rand(cpu.bounds.shape[0], generator=g)
    b = cpu.bounds
    b[r < 0.02, 0] = 0x7C00                                # centre.x = +Inf
    b[(r >= 0.02) & (r < 0.04), 5] = 0x7E00                # extent.y = NaN
    b[(r >= 0.04) & (r < 0.06), 6] = 0x7C00                # extent.z = +Inf
    b[(r >= 0.06) & (r < 0.08), 2] = torch.tensor(-1024, dtype=torch.int16)  # centre.z = 0xFC00 = -Inf
    gpu = cpu.to("cuda")
    light, mats, clip, zn, hpb = _vsm_inputs(67, 0.2)
    offs = clip.view(torch.int32).view(10, 19)[:, 16:18].numpy().copy()
    if huge_view is not None:  # (rows x and y scaled: an orthographic matrix stays one, its frustum is what it is; the point is the magnitude)
        mats = np.array(mats, dtype=np.float32).copy()
        mats[huge_view, 0::4] *= np.float32(2.0 ** 101)
        mats[huge_view, 1::4] *= np.float32(2.0 ** 101)
        clip = pack_clipmaps(mats, offs, zn)
    dirty_t = torch.ones(10, dtype=torch.int32)

    def camera(scene):
        cam = scene.cull_camera()
        for i in range(16):
            cam.projection_view[i] = float(mats[9][i])
        for i in range(3):
            cam.position[i] = float(-light[i])
        cam.near_clip = zn
        return cam

    cam = camera(cpu)
    mli, _ = oracle.cull_meshes(cpu, cam, L.CULL_TEST_FRUSTUM)
    h = oracle.make_hpb(hpb.data, 64, 64, 10, 7, hpb.level_offset)
    want_vis = oracle.cull_meshlets_hpb(cpu, cam, mli, clip, dirty_t, h)
    frame = PreparedFrame.create(gpu, expand=False)
    renderer.prepared_frame = frame
    ctx = CullGeometryContext(use_hpb=True, init_cull_meshes=True, cull_flags=L.CULL_TEST_FRUSTUM, cull_camera=camera(gpu), hpb_attachment=HpbAttachment(hpb.data.cuda(), 64, 64, 10, 7, hpb.level_offset),
                              vsm_clipmaps_buffer=clip.cuda(), vsm_clipmap_dirty_flags_buffer=dirty_t.cuda(), vsm_clipmap_count=10, stages=L.STAGE_MESHES | L.STAGE_MESHLETS)
    renderer.cull_geometry(ctx)
    c = renderer.read_counters(ctx)
    got_vis = frame.visible_meshlet_instances_indices_buffer[: c.cull_triangles_cmd_x].cpu()
    assert torch.equal(got_vis, want_vis)
    assert 0 < want_vis.numel() < mli.shape[0]


def test_use_hpb_argument_validation(renderer):
    from oxylus_amd.renderer import CullGeometryContext, PreparedFrame

    gpu = make_scene(SceneSpec(n_mesh_instances=2, meshlets_per_mesh=8, seed=1), "cuda")
    renderer.prepared_frame = PreparedFrame.create(gpu, expand=False)
    ctx = CullGeometryContext(use_hpb=True, init_cull_meshes=True, cull_flags=L.CULL_TEST_FRUSTUM, cull_camera=gpu.cull_camera())
    with pytest.raises(L.OxcError) as e:
        renderer.cull_geometry(ctx)
    assert e.value.status == L.OXC_INVALID_ARG


def test_cone_two_tier_near_threshold(renderer, oracle_lib):
    """The cone test has a cheap tier (hardware rsq/sqrt + margin) and the canonical IEEE tier for
    lanes inside the margin.  Force the rare tier: instances whose cone inequality is an exact tie or
    a few ulp off (lhs = 1 + j*ulp vs rhs = 1), mixed into waves with clearly decided meshlets."""
    spec = SceneSpec(n_mesh_instances=300, meshlets_per_mesh=5, seed=71, with_geometry=False)
    cpu = make_scene(spec, "cpu")
    b = cpu.bounds
    f16 = lambda *v: torch.tensor(v, dtype=torch.float32).to(torch.float16).view(torch.int16)  # noqa: E731
    # every 3rd meshlet of the first 256 instances: centre 0, extent (2,0,0) -> radius 1; axis (0,0,-1); cutoff 0
    for mi in range(256):
        for k in (0, 3):
            r = mi * 5 + k
            b[r, 0:3] = f16(0.0, 0.0, 0.0)
            b[r, 4:7] = f16(2.0, 0.0, 0.0)
            b[r, 3] = 0
            b[r, 7] = (129 & 0xFF) | (0 << 8)  # axis_z = -127, cutoff = 0
        j = mi - 128
        z = np.float32(-1.0)
        for _ in range(abs(j)):
            z = np.nextafter(z, np.float32(-2.0) if j > 0 else np.float32(0.0))
        t = torch.eye(4)
        t[3, 2] = float(z)  # column-major: translation z
        cpu.transforms[mi] = t.reshape(-1)
    gpu = cpu.to("cuda")
    want = oracle_frame(cpu, with_triangles=False)
    got = gpu_frame(renderer, gpu, with_triangles=False)
    assert_same(want, got, ["visible"])
    vis = set(want["visible"].tolist())
    # exact tie (j = 0) and lhs > rhs are culled, lhs < rhs is visible: both outcomes occur
    culled_probe = [mi * 5 for mi in range(256) if mi * 5 not in vis]
    assert 100 < len(culled_probe) < 256 and any(mi * 5 in vis for mi in range(256))


def test_meshlet_stage_one_million(renderer, oracle_lib):
    """BASELINE configs[1] at full size against the oracle (multi-threaded), unsorted bit-match."""
    import oracle

    spec = SceneSpec(n_mesh_instances=1000, meshlets_per_mesh=1000, seed=0x0A1DE5 + 2, with_geometry=False)
    gpu = make_scene(spec, "cuda")
    cpu = gpu.to("cpu")
    want = oracle.cull_meshlets(cpu, cpu.cull_camera(), cpu.meshlet_instances, nthreads=16)
    got = gpu_frame(renderer, gpu, with_triangles=False)
    assert np.array_equal(want.numpy(), got["visible"])
    assert 0.1 < want.numel() / 1e6 < 0.6


def test_two_pass_occlusion_ten_million_sampled(renderer, oracle_lib):
    """BASELINE configs[2] at full size (10M meshlets, 4096^2 HiZ, prior mask p = 0.3): the early and late decisions are
    per-meshlet independent, so the oracle is run on a 300K-instance PREFIX of the list (same HiZ, same mask words) and
    must reproduce exactly the part of the GPU's early / late lists and mask that falls into that prefix; the lists of the
    whole frame must be ascending and their sizes consistent with the counters."""
    import oracle
    from oxylus_amd.renderer import CullGeometryContext, PreparedFrame

    spec = SceneSpec(n_mesh_instances=10000, meshlets_per_mesh=1000, seed=0x0A1DE5 + 3, with_geometry=False)
    gpu = make_scene(spec, "cuda")
    N, P = gpu.n_meshlet_instances, 300_000
    depth = make_depth(8192, 8192, 64, seed=3, device="cuda")
    hiz = ImageAttachment.hiz(4096, 4096, "cuda")
    renderer.generate_hiz(MainGeometryContext(ImageAttachment.depth(depth), hiz))
    g = torch.Generator(device="cuda").manual_seed(5)
    words = (N + 31) // 32
    mask0 = torch.randint(-2**31, 2**31 - 1, (words,), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
    mask0 &= torch.randint(-2**31, 2**31 - 1, (words,), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)  # p ~ 0.25
    frame = PreparedFrame.create(gpu, with_triangles=False)
    frame.meshlet_instance_visibility_mask_buffer.copy_(mask0)
    renderer.prepared_frame = frame
    ctx = CullGeometryContext(use_hiz=True, init_cull_meshes=False, cull_flags=L.CULL_TEST_ALL, cull_camera=gpu.cull_camera(), hiz_attachment=hiz,
                              stages=L.STAGE_MESHLETS)
    renderer.seed_meshlet_instances(ctx, N)
    renderer.cull_geometry(ctx)
    c1 = renderer.read_counters(ctx)
    early = frame.visible_meshlet_instances_indices_buffer[: c1.cull_triangles_cmd_x].clone()
    ctx.cull_flags = L.CULL_TEST_ALL | L.CULL_LATE_PASS
    renderer.cull_geometry(ctx)
    c2 = renderer.read_counters(ctx)
    late = frame.visible_meshlet_instances_indices_buffer[c2.early_visible_meshlet_instances: c2.early_visible_meshlet_instances + c2.cull_triangles_cmd_x].clone()
    assert c2.early_visible_meshlet_instances == early.numel() and c2.late_visible_meshlet_instances == late.numel()
    for lst in (early, late):
        assert bool((lst[1:] > lst[:-1]).all())  # ascending, no duplicates
    assert early.numel() > 0 and late.numel() > 0 and not bool(torch.isin(late[:100000], early).any())
    # oracle on the prefix
    cpu = gpu.to("cpu")
    hiz_host = hiz.data.cpu().numpy()  # kept alive: make_hiz stores the pointer
    hz = oracle.make_hiz(hiz_host, 4096, 4096, hiz.levels, hiz.level_offset)
    mli = cpu.meshlet_instances[:P].contiguous()
    mask = mask0.cpu().clone()
    v = oracle.Visibility(P, 0, 0)
    out = torch.zeros(P, dtype=torch.int32)
    n_e = oracle.cull_meshlets_hiz(cpu, cpu.cull_camera(), mli, L.CULL_TEST_ALL, hz, v, mask, out)
    want_early = out[:n_e].clone()
    n_l = oracle.cull_meshlets_hiz(cpu, cpu.cull_camera(), mli, L.CULL_TEST_ALL | L.CULL_LATE_PASS, hz, v, mask, out)
    want_late = out[v.early: v.early + n_l].clone()
    e_cpu, l_cpu = early.cpu(), late.cpu()
    assert torch.equal(e_cpu[e_cpu < P], want_early) and torch.equal(l_cpu[l_cpu < P], want_late)
    # mask words wholly inside the prefix (mask index = visibility offset + meshlet index = list position here)
    wp = P // 32
    assert torch.equal(frame.meshlet_instance_visibility_mask_buffer[:wp].cpu(), mask[:wp])


@pytest.mark.parametrize("run_cull_meshes", [True, False])
def test_cull_geometry_batch_equals_individual_calls(renderer, oracle_lib, run_cull_meshes):
    """oxc_cull_geometry_batch: several independent frames per launch (grid.y = batch element) must give
    exactly what one call per frame gives."""
    from oxylus_amd.renderer import CullGeometryContext, PreparedFrame

    specs = [SceneSpec(n_mesh_instances=40, meshlets_per_mesh=90, lod_count=3, seed=81),
             SceneSpec(n_mesh_instances=7, meshlets_per_mesh=500, lod_count=3, seed=82, ragged=True),
             SceneSpec(n_mesh_instances=120, meshlets_per_mesh=11, lod_count=3, seed=83),
             SceneSpec(n_mesh_instances=1, meshlets_per_mesh=3000, lod_count=3, seed=84)]
    pairs = [_pair(sp) for sp in specs]
    wants = [oracle_frame(cpu, run_cull_meshes=run_cull_meshes) for cpu, _ in pairs]
    frames, ctxs = [], []
    for _, gpu in pairs:
        fr = PreparedFrame.create(gpu, expand=not run_cull_meshes)
        cx = CullGeometryContext(init_cull_meshes=run_cull_meshes, cull_flags=L.CULL_TEST_ALL, cull_camera=gpu.cull_camera())
        if not run_cull_meshes:
            renderer.prepared_frame = fr
            renderer.seed_meshlet_instances(cx, gpu.n_meshlet_instances)
        frames.append(fr)
        ctxs.append(cx)
    renderer.cull_geometry_batch(frames, ctxs)
    for want, fr, cx, (_, gpu) in zip(wants, frames, ctxs, pairs):
        c = renderer.read_counters(cx)
        assert c.total_visible_meshlet_instances == want["total"]
        got_vis = fr.visible_meshlet_instances_indices_buffer[: c.cull_triangles_cmd_x].cpu().numpy()
        got_idx = fr.reordered_indices_buffer[: c.draw_index_count].cpu().numpy()
        assert np.array_equal(got_vis, want["visible"])
        assert np.array_equal(got_idx, want["indices"])
        if run_cull_meshes:
            assert np.array_equal(fr.meshlet_instances_buffer[: want["total"]].cpu().numpy(), want["meshlet_instances"])
            assert np.array_equal(gpu.mesh_instances[:, 1].cpu().numpy(), want["lod_index"])


def test_stage_subsets_and_argument_validation(renderer, oracle_lib):
    """`stages` runs a prefix of the pipeline; bad arguments come back as OXC_INVALID_ARG with a message."""
    from oxylus_amd.renderer import CullGeometryContext, PreparedFrame

    spec = SceneSpec(n_mesh_instances=20, meshlets_per_mesh=50, lod_count=2, seed=91)
    cpu, gpu = _pair(spec)
    want = oracle_frame(cpu, run_cull_meshes=True)
    # meshes only
    frame = PreparedFrame.create(gpu, expand=False)
    renderer.prepared_frame = frame
    ctx = CullGeometryContext(init_cull_meshes=True, cull_flags=L.CULL_TEST_ALL, cull_camera=gpu.cull_camera(), stages=L.STAGE_MESHES)
    renderer.cull_geometry(ctx)
    c = renderer.read_counters(ctx)
    assert c.total_visible_meshlet_instances == want["total"] and c.cull_meshlets_cmd_x == want["cull_meshlets_cmd_x"]
    assert c.cull_triangles_cmd_x == 0 and c.draw_index_count == 0
    assert np.array_equal(frame.meshlet_instances_buffer[: want["total"]].cpu().numpy(), want["meshlet_instances"])
    # then meshlets + triangles as a later call of the same sequence
    ctx.init_cull_meshes = False
    ctx.stages = L.STAGE_MESHLETS | L.STAGE_TRIANGLES
    renderer.cull_geometry(ctx)
    c = renderer.read_counters(ctx)
    assert np.array_equal(frame.visible_meshlet_instances_indices_buffer[: c.cull_triangles_cmd_x].cpu().numpy(), want["visible"])
    assert np.array_equal(frame.reordered_indices_buffer[: c.draw_index_count].cpu().numpy(), want["indices"])

    def expect_invalid(mutate):
        fr = PreparedFrame.create(gpu, expand=True)
        cx = CullGeometryContext(init_cull_meshes=True, cull_flags=L.CULL_TEST_ALL, cull_camera=gpu.cull_camera())
        mutate(fr, cx)
        renderer.prepared_frame = fr
        with pytest.raises(L.OxcError) as e:
            renderer.cull_geometry(cx)
        assert e.value.status == L.OXC_INVALID_ARG and len(str(e.value)) > 20

    def too_small_reordered(fr, cx):
        fr.reordered_indices_buffer = fr.reordered_indices_buffer[:100]

    def wrong_instance_count(fr, cx):
        cx.cull_camera.mesh_instance_count += 1

    def hiz_without_image(fr, cx):
        cx.use_hiz = True

    def late_without_sequence(fr, cx):
        cx.init_cull_meshes = False

    for m in (too_small_reordered, wrong_instance_count, hiz_without_image, late_without_sequence):
        expect_invalid(m)


@pytest.mark.parametrize("count", [3, 8, 9, 16, 18], ids=["3", "8-one-full-piece", "9-two-pieces", "16-max", "18-falls-back"])
def test_batch_sizes_beyond_one_kernarg_piece(renderer, oracle_lib, count):
    """Up to 16 elements are fused (their argument cores are handed over by prepare launches of <= 8 elements each, which
    rebuild the per-stage argument blocks on the device); more than 16 (or a HiZ element) is processed one call at a time --
    always with the same results."""
    from oxylus_amd.renderer import CullGeometryContext, PreparedFrame

    pairs = [_pair(SceneSpec(n_mesh_instances=6, meshlets_per_mesh=40 + 7 * i, seed=100 + i)) for i in range(count)]
    wants = [oracle_frame(cpu) for cpu, _ in pairs]
    frames, ctxs = [], []
    for _, gpu in pairs:
        fr = PreparedFrame.create(gpu)
        cx = CullGeometryContext(init_cull_meshes=False, cull_flags=L.CULL_TEST_ALL, cull_camera=gpu.cull_camera())
        renderer.prepared_frame = fr
        renderer.seed_meshlet_instances(cx, gpu.n_meshlet_instances)
        frames.append(fr)
        ctxs.append(cx)
    renderer.cull_geometry_batch(frames, ctxs)
    for want, fr, cx in zip(wants, frames, ctxs):
        c = renderer.read_counters(cx)
        assert np.array_equal(fr.visible_meshlet_instances_indices_buffer[: c.cull_triangles_cmd_x].cpu().numpy(), want["visible"])
        assert np.array_equal(fr.reordered_indices_buffer[: c.draw_index_count].cpu().numpy(), want["indices"])


@pytest.mark.parametrize("spec", [SceneSpec(n_mesh_instances=12, meshlets_per_mesh=70, tris_per_meshlet=124, seed=111),
                                  SceneSpec(n_mesh_instances=5, meshlets_per_mesh=33, tris_per_meshlet=128, seed=112, ragged=True)],
                         ids=["124tris", "ragged<=128"])
def test_wide_triangle_index_extension(renderer, oracle_lib, spec):
    """SURVEY A.7 extension: meshlets of up to 128 triangles with the (id << 9) | (3t+k) index.  No
    reference behaviour exists for it (the reference's 8-bit corner field stops at 85 triangles); parity is
    against the oracle's statement of the same rule."""
    import oracle
    from oxylus_amd.renderer import CullGeometryContext, PreparedFrame

    cpu, gpu = _pair(spec)
    cam = cpu.cull_camera()
    want_vis = oracle.cull_meshlets(cpu, cam, cpu.meshlet_instances)
    want_idx = oracle.cull_triangles(cpu, cam, cpu.meshlet_instances, want_vis, 0, want_vis.numel(), wide=True)
    frame = PreparedFrame.create(gpu, max_tris=128)
    renderer.prepared_frame = frame
    ctx = CullGeometryContext(init_cull_meshes=False, cull_flags=L.CULL_TEST_ALL, cull_camera=gpu.cull_camera(), wide_triangle_index=True)
    renderer.seed_meshlet_instances(ctx, gpu.n_meshlet_instances)
    renderer.cull_geometry(ctx)
    c = renderer.read_counters(ctx)
    assert torch.equal(frame.visible_meshlet_instances_indices_buffer[: c.cull_triangles_cmd_x].cpu(), want_vis)
    got = frame.reordered_indices_buffer[: c.draw_index_count].cpu()
    assert torch.equal(got, want_idx)
    assert int((got & 0x1FF).max()) > 255  # corners beyond the reference's 8-bit field are actually used


def test_triangle_stage_empty_and_oversized_meshlets(renderer, oracle_lib):
    """Meshlets with triangle_count == 0 emit nothing and must not make the straight-line triangle kernel read the
    mesh through their (possibly dangling) offsets; triangle_count > 64 is cut to the kernel's 64 threads
    (defines.slang:9-11, cull_triangles.slang:44).  Also an all-empty visible set and a visible count that is not a
    multiple of the 64-slot block."""
    spec = SceneSpec(n_mesh_instances=9, meshlets_per_mesh=75, seed=321)
    cpu = make_scene(spec, "cpu")
    g = torch.Generator().manual_seed(5)
    r = torch.rand(cpu.meshlets.shape[0], generator=g)
    cpu.meshlets[r < 0.2, 3] = 0
    cpu.meshlets[(r >= 0.2) & (r < 0.3), 3] = 200
    # a dangling offset on an empty meshlet: nothing may be read through it
    empty = torch.nonzero(r < 0.2).flatten()
    cpu.meshlets[empty[0], 0] = 0x3FFFFFFF
    cpu.meshlets[empty[0], 1] = 0x3FFFFFFC
    gpu = cpu.to("cuda")
    want = oracle_frame(cpu)
    got = gpu_frame(renderer, gpu)
    assert_same(want, got, ["total", "visible", "indices"])
    assert len(want["indices"]) > 0
    # every meshlet empty
    cpu.meshlets[:, 3] = 0
    gpu = cpu.to("cuda")
    want = oracle_frame(cpu)
    got = gpu_frame(renderer, gpu)
    assert_same(want, got, ["total", "visible", "indices"])
    assert len(want["indices"]) == 0 and len(want["visible"]) > 0


@pytest.mark.parametrize("m,k", [(3000, 5), (16384, 2), (20000, 2)], ids=["3-tiles", "16-tiles", "tile-loop"])
def test_cull_meshes_scan_paths(renderer, oracle_lib, m, k):
    """k_scan_mesh_counts: the all-in-registers path (<= 16K mesh instances) and the tile loop beyond it."""
    spec = SceneSpec(n_mesh_instances=m, meshlets_per_mesh=k, lod_count=2, seed=900 + m, with_geometry=False, scene_depth=400.0)
    cpu, gpu = _pair(spec)
    want = oracle_frame(cpu, run_cull_meshes=True, with_triangles=False)
    got = gpu_frame(renderer, gpu, run_cull_meshes=True, with_triangles=False)
    assert_same(want, got, ["total", "cull_meshlets_cmd_x", "lod_index", "meshlet_instances", "visible"])
    assert 0 < want["total"] < m * k


def test_project_aabb_matches_ieee_division_bit_for_bit(renderer, oracle_lib):
    """The device's shared-reciprocal division path (oxcull_device.hpp project_aabb) against the oracle's IEEE divisions:
    all six outputs bit-identical, for boxes inside the exponent window (fast path), and for waves that contain boxes
    with exact-zero numerators, astronomically large coordinates or a tiny z (IEEE fallback for that wave)."""
    import oracle
    from oxylus_amd.synth import perspective_reversed_z

    g = torch.Generator().manual_seed(77)
    n = 1 << 14
    centers = torch.cat([(torch.rand((n, 2), generator=g) - 0.5) * 400.0, -torch.rand((n, 1), generator=g) * 900.0 - 0.2], 1)
    extents = torch.exp(torch.rand((n, 3), generator=g) * 6.0 - 3.0)
    boxes = torch.cat([centers, extents], 1).contiguous()
    # special waves (64 consecutive boxes share a wave): exact zeros, huge values, tiny z numerators
    boxes[64 * 3 + 5] = torch.tensor([0.0, 0.0, -10.0, 0.0, 0.0, 0.0])          # corner x = y = 0 exactly, zero extent
    boxes[64 * 7 + 9] = torch.tensor([3.0e30, 1.0, -50.0, 1.0, 1.0, 1.0])         # |numerator| beyond 2^60
    boxes[64 * 11 + 1] = torch.tensor([1.0, 2.0, -1.0e-25, 1.0e-26, 1.0e-26, 1.0e-26])  # crosses / hugs the near plane
    pv = perspective_reversed_z(60.0, 1.0, 0.1, 1000.0)
    # (the device entry evaluates both forms of project_aabb -- the general one and the one with the w == 1 short cut the clipmap views
    #  take -- and marks a box where they disagree with 2 in the seventh float; an orthographic matrix takes the short cut except in
    #  the waves with the astronomically large / non-finite boxes, the scaled one never does)
    ortho = [0.01, 0, 0, 0, 0, 0.02, 0, 0, 0, 0, 0.001, 0, 0.1, -0.2, 0.5, 1.0]
    rot = [0.006, 0.008, 0.0005, 0, -0.016, 0.012, 0.0003, 0, 0.001, -0.002, 0.001, 0, 0.1, -0.2, 0.5, 1.0]
    mats = {"perspective": pv.tolist(), "orthographic-w=1": ortho, "orthographic-rotated": rot, "orthographic-w=2": [2.0 * x for x in ortho]}
    for name, m in mats.items():
        near = 0.1 if name == "perspective" else 0.01
        if name == "orthographic-rotated":
            boxes = boxes.clone()
            boxes[64 * 20 + 3] = torch.tensor([1.0, 2.0, -3.0, float("inf"), 1.0, 1.0])
            boxes[64 * 21 + 4] = torch.tensor([3.0e38, 3.0e38, -3.0e38, 3.0e38, 3.0e38, 3.0e38])
            boxes[64 * 22 + 5] = torch.tensor([float("nan"), 2.0, -3.0, 1.0, 1.0, 1.0])
        got = renderer.debug_project_aabb(m, near, boxes.cuda()).cpu().numpy()
        want = np.zeros((n, 7), dtype=np.float32)
        for i in range(n):
            r = oracle.project_aabb(m, near, boxes[i, :3].numpy(), boxes[i, 3:].numpy())
            if r is not None:
                want[i, :6] = r
                want[i, 6] = 1.0
        assert np.array_equal(got[:, 6], want[:, 6]), name
        ok = want[:, 6] == 1.0
        gu, wu = got[ok][:, [0, 1, 3, 4, 5]].view(np.uint32), want[ok][:, [0, 1, 3, 4, 5]].view(np.uint32)
        assert np.array_equal(gu, wu), f"{name}: {int((gu != wu).any(1).sum())} boxes differ"
        assert ok.sum() > n // 4


def test_comm_entry_points_single_rank(oracle_lib):
    """oxc_comm_* / oxc_exchange_counts / oxc_broadcast_hiz on a world of one: RCCL is loaded, a communicator
    comes up on the stream, the all-gather returns this rank's counters, the broadcast leaves the root's pyramid intact."""
    from oxylus_amd.renderer import ImageAttachment, RendererInstance

    r = RendererInstance(0)
    uid = r.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    r.comm_init(uid, 0, 1)
    mine = torch.tensor([11, 22, 33, 44], dtype=torch.int32, device="cuda")
    got = r.exchange_counts(mine)
    hiz = ImageAttachment.hiz(64, 64, "cuda")
    hiz.data.copy_(torch.arange(hiz.data.numel(), dtype=torch.float32))
    r.broadcast_hiz(hiz, 0)
    torch.cuda.synchronize()
    assert got.cpu().tolist() == [[11, 22, 33, 44]]
    assert torch.equal(hiz.data.cpu(), torch.arange(hiz.data.numel(), dtype=torch.float32))
    from oxylus_amd.lib import OxcError
    with pytest.raises(OxcError):
        r.comm_init(uid, 0, 1)  # already initialised
    r.comm_destroy()
    with pytest.raises(OxcError):
        r.exchange_counts(mine)


def test_maximum_instance_count_of_the_packed_index(renderer, oracle_lib):
    """The reference's packed index holds a 24-bit meshlet-instance id (visbuffer.slang:9-14): the largest frame the
    triangle stage accepts has 2^24 instances.  Just below the limit the ids in the top byte range must come out right
    (compared with the oracle on the tail of the visible list); above it the call is refused."""
    import oracle
    from oxylus_amd.lib import OxcError
    from oxylus_amd.renderer import CullGeometryContext, PreparedFrame

    K = 1000
    spec = SceneSpec(n_mesh_instances=16777, meshlets_per_mesh=K, share_meshes=2, seed=501, scene_depth=900.0)   # 16 777 000 < 2^24
    cpu = make_scene(spec, "cpu")
    gpu = cpu.to("cuda")
    N = cpu.n_meshlet_instances
    cam = cpu.cull_camera()
    want_vis = oracle.cull_meshlets(cpu, cam, cpu.meshlet_instances, nthreads=16)
    frame = PreparedFrame.create(gpu)
    renderer.prepared_frame = frame
    ctx = CullGeometryContext(init_cull_meshes=False, cull_flags=L.CULL_TEST_ALL, cull_camera=gpu.cull_camera())
    renderer.seed_meshlet_instances(ctx, N)
    renderer.cull_geometry(ctx)
    c = renderer.read_counters(ctx)
    got_vis = frame.visible_meshlet_instances_indices_buffer[: c.cull_triangles_cmd_x].cpu()
    assert torch.equal(got_vis, want_vis)
    assert int(want_vis.max()) >= (1 << 23)                 # ids with the top bit of the 24-bit field set are in play
    # triangles of the last 64 visible meshlets (largest ids): their packed indices are the tail of the GPU list
    tail = 64
    want_tail = oracle.cull_triangles(cpu, cam, cpu.meshlet_instances, want_vis, want_vis.numel() - tail, tail)
    got_idx = frame.reordered_indices_buffer[c.draw_index_count - want_tail.numel(): c.draw_index_count].cpu()
    assert torch.equal(got_idx, want_tail)
    assert int((got_idx.to(torch.int64) & 0xFFFFFFFF).max() >> 8) == int(want_vis[-tail:].max()) or want_tail.numel() == 0
    del frame, gpu
    # one mesh instance more: 16 778 000 > 2^24 -> refused when the triangle stage runs, accepted without it
    spec2 = SceneSpec(n_mesh_instances=16778, meshlets_per_mesh=K, share_meshes=2, seed=501, with_geometry=False, scene_depth=900.0)
    big = make_scene(spec2, "cuda")
    frame = PreparedFrame.create(big, with_triangles=False)
    renderer.prepared_frame = frame
    ctx = CullGeometryContext(init_cull_meshes=False, cull_flags=L.CULL_TEST_ALL, cull_camera=big.cull_camera(), stages=L.STAGE_ALL)
    renderer.seed_meshlet_instances(ctx, big.n_meshlet_instances)
    with pytest.raises(OxcError):
        renderer.cull_geometry(ctx)
    ctx.stages = L.STAGE_MESHLETS
    renderer.cull_geometry(ctx)
    assert renderer.read_counters(ctx).total_visible_meshlet_instances == big.n_meshlet_instances
