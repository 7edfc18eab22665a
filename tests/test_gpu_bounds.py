"""GPU parity of the meshlet bounds producer (oxc_build_meshlet_bounds) against the oracle:
MeshletBounds records, quantised positions and the mesh AABB must be byte-identical."""
import numpy as np
import pytest
import torch

from oxylus_amd.synth import build_meshlets_simple, make_mesh

pytestmark = pytest.mark.gpu


def _both(renderer, pos, meshlets, vidx, micro):
    import oracle

    want = oracle.build_meshlet_bounds(pos, meshlets, vidx, micro)
    got = renderer.build_meshlet_bounds(pos.cuda(), meshlets.cuda(), vidx.cuda(), micro.cuda())
    torch.cuda.synchronize()
    return want, tuple(t.cpu() for t in got)


@pytest.mark.parametrize("kind,n,max_tris", [("sphere", 24, 64), ("terrain", 24, 64), ("soup", 12, 64), ("plane0", 16, 64),
                                             ("sphere", 40, 124), ("terrain", 40, 200)],
                         ids=["sphere", "terrain+degenerate", "soup", "signed-zero-plane", "sphere-124tris", "terrain-200tris"])
def test_bounds_match_oracle(renderer, oracle_lib, kind, n, max_tris):
    pos, tris = make_mesh(kind, n=n, seed=11)
    meshlets, vidx, micro = build_meshlets_simple(tris, max_vertices=255 if max_tris > 64 else 64, max_triangles=max_tris)
    (wb, wm, wq), (gb, gm, gq) = _both(renderer, pos, meshlets, vidx, micro)
    assert torch.equal(wb, gb)
    assert torch.equal(wq, gq)
    assert np.array_equal(wm.numpy().view(np.uint32), gm.numpy().view(np.uint32))
    if kind in ("sphere", "terrain"):
        cut = (wb.numpy().view(np.uint16)[:, 7] >> 8).astype(np.int8)
        assert (cut < 127).any()  # real cones, not only the degenerate answer


def test_bounds_empty_and_tiny_inputs(renderer, oracle_lib):
    pos, tris = make_mesh("sphere", n=8, seed=2)
    meshlets, vidx, micro = build_meshlets_simple(tris)
    meshlets[1, 3] = 0  # a meshlet without triangles: FLT_MAX fold, cone data zero
    (wb, wm, wq), (gb, gm, gq) = _both(renderer, pos, meshlets, vidx, micro)
    assert torch.equal(wb, gb) and torch.equal(wq, gq)
    assert np.array_equal(wm.numpy().view(np.uint32), gm.numpy().view(np.uint32))
    # a single one-triangle meshlet
    one = torch.tensor([[0, 0, 3, 1]], dtype=torch.int32)
    (wb, wm, _), (gb, gm, _) = _both(renderer, pos, one, vidx[:3].contiguous(), torch.tensor([0, 1, 2, 0], dtype=torch.uint8))
    assert torch.equal(wb, gb) and np.array_equal(wm.numpy().view(np.uint32), gm.numpy().view(np.uint32))


def test_produced_bounds_feed_the_cull_path(renderer, oracle_lib):
    """The producer's records are what the cull kernels decode: every box must contain its vertices after the
    half round trip of the centre/extent (up to the half rounding of centre and extent), and the decoded cone
    axis is the s8 the producer wrote."""
    pos, tris = make_mesh("terrain", n=24, seed=4)
    meshlets, vidx, micro = build_meshlets_simple(tris)
    gb, _, _ = renderer.build_meshlet_bounds(pos.cuda(), meshlets.cuda(), vidx.cuda(), micro.cuda())
    dec = renderer.debug_decode_bounds(gb).cpu().numpy()
    P = pos.numpy()
    for m, (vo, to, vc, tc) in enumerate(meshlets.tolist()):
        ids = [int(vidx[vo + int(micro[to + i])]) for i in range(3 * tc)]
        lo, hi = P[ids].min(0), P[ids].max(0)
        c, e = dec[m, 0:3], dec[m, 3:6]
        tol = 2.0 ** -10 * (np.abs(c) + e + 1e-3) * 2
        assert np.all(c - e * 0.5 <= lo + tol) and np.all(c + e * 0.5 >= hi - tol)


@pytest.mark.parametrize("kind", ["sphere", "terrain"])
def test_real_mesh_through_producer_and_cull_pipeline(renderer, oracle_lib, kind):
    """Asset path -> cull path on a real mesh: bounds / positions produced on the GPU (== the checker's, asserted
    above) feed cull_meshes + cull_meshlets + cull_triangles; outputs must match the oracle run on the checker's arrays."""
    import oracle
    from oxylus_amd.synth import make_scene_from_mesh
    from util import assert_same, gpu_frame, oracle_frame

    pos, tris = make_mesh(kind, n=28, seed=21)
    meshlets, vidx, micro = build_meshlets_simple(tris)
    wb, wm, wq = oracle.build_meshlet_bounds(pos, meshlets, vidx, micro)
    gb, gm, gq = renderer.build_meshlet_bounds(pos.cuda(), meshlets.cuda(), vidx.cuda(), micro.cuda())
    cpu = make_scene_from_mesh(40, wb, meshlets, micro, vidx, wq, wm, seed=77, device="cpu", scene_depth=60.0)
    gpu = cpu.to("cuda")  # same instance placement (the torch CPU and GPU generators differ) ...
    gpu.bounds, gpu.positions = gb.contiguous().clone(), gq.contiguous().clone()  # ... with the GPU-produced records
    gpu.meshes.view(torch.int32)[0, 10:16] = gm.view(torch.int32)
    gpu.bind()
    want = oracle_frame(cpu, run_cull_meshes=True)
    got = gpu_frame(renderer, gpu, run_cull_meshes=True)
    assert_same(want, got, ["total", "visible", "indices"])
    assert 0 < len(want["visible"]) < want["total"]     # both the frustum and the produced cones reject something
    assert len(want["indices"]) > 0
