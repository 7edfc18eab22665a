"""Terrain patch cull (oxc_cull_terrain) against the oracle's restatement of
passes/terrain_cull.slang:17-83: emitted patch list (ascending), indirect command and mask words byte-identical.
The reference has no tests for this pass; the small CPU cases below are hand-checked."""
import numpy as np
import pytest
import torch

from oxylus_amd import lib as L
from oxylus_amd.lib import CullCamera
from oxylus_amd.synth import make_depth, perspective_reversed_z


def _camera(eye=(0.0, 40.0, 0.0), near=0.1, far=2000.0):
    """Camera at `eye` looking down -Z (view = translate(-eye)); reversed-Z projection as everywhere else."""
    proj = perspective_reversed_z(60.0, 1.0, near, far).view(4, 4)  # [col][row]
    view = torch.eye(4)
    view[3, 0:3] = -torch.tensor(eye)  # column 3 = translation
    pv = (proj.t() @ view.t()).t().contiguous()  # column-major product proj * view
    cam = CullCamera()
    for i, v in enumerate(pv.flatten().tolist()):
        cam.projection_view[i] = v
    for i in range(3):
        cam.position[i] = eye[i]
    cam.near_clip = near
    cam.resolution[0] = cam.resolution[1] = 1024.0
    cam.acceptable_lod_error = 2.0
    return cam


def _terrain(pcx, pcy, seed):
    g = torch.Generator().manual_seed(seed)
    lo = torch.rand((pcy, pcx), generator=g) * 0.6
    hi = lo + torch.rand((pcy, pcx), generator=g) * 0.4
    flat = torch.rand((pcy, pcx), generator=g) < 0.1
    hi = torch.where(flat, lo, hi)  # flat patches: extent.y falls back to 1e-3
    return torch.stack([lo, hi], -1).contiguous()


def _mask(total, seed, p):
    g = torch.Generator().manual_seed(seed)
    words = (total + 31) // 32
    bits = (torch.rand((words, 32), generator=g) < p).to(torch.int64)
    return (bits << torch.arange(32)).sum(1).to(torch.int32)


def test_oracle_frustum_only_keeps_the_patches_in_front(oracle_lib):
    import oracle

    # 4 x 2 patches on [-40, 40] x [-100, -20] (x, z): all in front of a camera at the origin looking down -z
    # except that TestFrustum alone (early pass) only keeps previously visible ones (mask), terrain_cull.slang:47
    mm = torch.zeros((2, 4, 2))
    mm[..., 1] = 0.5
    cam = _camera(eye=(0.0, 10.0, 0.0))
    mask = torch.tensor([0b10110101], dtype=torch.int32)
    got = oracle.cull_terrain([-40.0, -100.0], [80.0, 80.0], (4, 2), 0.0, 10.0, mm, cam, L.CULL_TEST_FRUSTUM, None, mask)
    assert got.tolist() == [0, 2, 4, 5, 7]     # exactly the mask's patches: everything is inside the frustum
    assert mask.tolist() == [0b10110101]       # no TestOcclusion / LatePass: the mask is not written (terrain_cull.slang:60)
    behind = oracle.cull_terrain([-40.0, 20.0], [80.0, 80.0], (4, 2), 0.0, 10.0, mm, cam, L.CULL_TEST_FRUSTUM, None, mask)
    assert behind.tolist() == []               # z in [20, 100]: behind the camera


@pytest.mark.gpu
@pytest.mark.parametrize("pcx,pcy,flags", [(64, 64, L.CULL_TEST_FRUSTUM), (64, 64, L.CULL_TEST_FRUSTUM | L.CULL_TEST_OCCLUSION),
                                           (37, 29, L.CULL_TEST_FRUSTUM | L.CULL_TEST_OCCLUSION), (1, 1, L.CULL_TEST_ALL), (130, 9, L.CULL_TEST_ALL),
                                           (32, 32, L.CULL_TEST_ALL), (333, 257, L.CULL_TEST_ALL)])  # 32 x 32: exactly one block (appends in the test kernel); 333 x 257: 84 blocks
def test_gpu_terrain_cull_early_then_late(renderer, oracle_lib, pcx, pcy, flags):
    import oracle
    from oxylus_amd.renderer import ImageAttachment
    from util import oracle_hiz

    total = pcx * pcy
    mm = _terrain(pcx, pcy, 100 + pcx)
    cam = _camera()
    depth = make_depth(512, 512, 40, seed=9)
    data, levels, offs = oracle_hiz(depth, 256, 256)
    hz = oracle.make_hiz(data.numpy(), 256, 256, levels, offs)
    att = ImageAttachment.hiz(256, 256, "cuda")
    att.data.copy_(data.cuda())
    wmin, wsize, base_h, hscale = [-300.0, -700.0], [600.0, 650.0], -5.0, 60.0
    mask_cpu = _mask(total, 7, 0.35)
    mask_gpu = mask_cpu.clone().cuda()
    for pass_flags in (flags, flags | L.CULL_LATE_PASS):
        want = oracle.cull_terrain(wmin, wsize, (pcx, pcy), base_h, hscale, mm, cam, pass_flags, hz, mask_cpu)
        got, cmd = renderer.cull_terrain(pass_flags, cam, wmin, wsize, (pcx, pcy), base_h, hscale, mm.cuda(), mask_gpu, hiz=att)
        assert cmd == [4, want.numel(), 0, 0]
        assert torch.equal(got.cpu(), want)
        assert torch.equal(mask_gpu.cpu(), mask_cpu)
    if total > 1000:
        assert 0 < want.numel() < total


@pytest.mark.gpu
def test_gpu_terrain_argument_validation(renderer):
    from oxylus_amd.lib import OxcError

    cam = _camera()
    mm = torch.zeros((4, 4, 2), device="cuda")
    with pytest.raises(OxcError):  # occlusion without a HiZ
        renderer.cull_terrain(L.CULL_TEST_ALL, cam, [0, 0], [1, 1], (4, 4), 0.0, 1.0, mm, torch.zeros(1, dtype=torch.int32, device="cuda"))
    with pytest.raises(OxcError):  # mask too small
        renderer.cull_terrain(L.CULL_TEST_FRUSTUM, cam, [0, 0], [1, 1], (8, 8), 0.0, 1.0, torch.zeros((8, 8, 2), device="cuda"),
                              torch.zeros(1, dtype=torch.int32, device="cuda"))
