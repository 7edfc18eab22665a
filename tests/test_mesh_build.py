"""Asset-side clusteriser (oxc_mesh_build_*, the per-LOD loop of AssetManager_GLTF.cpp:599-682).  meshoptimizer is not
available (third-party, outside the reference tree), so there is no output to match triangle for triangle; what is pinned is the
contract downstream code relies on: format, validity, the reference's loop rules, determinism -- and, on the GPU, that the result
flows through the bounds producer and the whole cull pipeline to the checker's bytes."""
import numpy as np
import pytest
import torch

import oracle
from oxylus_amd import lib as L
from oxylus_amd.mesh_build import build_mesh_lods, make_scene_from_lods, reorder_vertices, vertex_fetch_remap
from oxylus_amd.synth import make_mesh


def _triangles_of(lod):
    m, out = lod["meshlets"], []
    for k in range(m.shape[0]):
        vo, to, vc, tc = m[k].tolist()
        loc = lod["micro"][to:to + tc * 3].view(-1, 3).long()
        assert loc.numel() == 0 or int(loc.max()) < vc
        out.append(lod["vidx"][vo:vo + vc].long()[loc])
    return torch.cat(out) if out else torch.zeros((0, 3), dtype=torch.long)


def _canon(tris):  # multiset of triangles, rotation-normalised (winding kept)
    t = tris.numpy()
    r = np.argmin(t, axis=1)
    rolled = np.stack([t[np.arange(len(t)), (r + k) % 3] for k in range(3)], 1)
    return sorted(map(tuple, rolled.tolist()))


def _border_edges(idx):
    t = idx.view(-1, 3).numpy()
    t = t[(t[:, 0] != t[:, 1]) & (t[:, 1] != t[:, 2]) & (t[:, 0] != t[:, 2])]  # (LOD 0 keeps the input's degenerate triangles: no area, no edges)
    e = np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]])
    e.sort(axis=1)
    u, c = np.unique(e, axis=0, return_counts=True)
    return set(map(tuple, u[c == 1].tolist()))


@pytest.mark.parametrize("kind,n", [("sphere", 24), ("terrain", 40), ("soup", 12)])
def test_every_lod_is_a_valid_clustering(liboxcull, kind, n):
    pos, tris = make_mesh(kind, n=n, seed=1)
    lods = build_mesh_lods(pos, tris)
    assert 1 <= len(lods) <= 8
    for lod in lods:
        m = lod["meshlets"]
        assert m.shape[0] > 0 and int(m[:, 2].max()) <= 64 and int(m[:, 3].max()) <= 64 and int(m[:, 2].min()) >= 3
        assert bool((m[:, 1] % 4 == 0).all()) and lod["micro"].numel() % 4 == 0            # 4-byte aligned micro-index runs
        assert m[:, 0].tolist() == (torch.cumsum(m[:, 2], 0) - m[:, 2]).tolist()            # vertex runs packed back to back
        assert int(m[-1, 0] + m[-1, 2]) == lod["vidx"].numel()
        # every triangle of the LOD (bar the degenerate ones: no area) in exactly one meshlet, winding preserved; no vertex listed twice in a meshlet
        li = lod["indices"].view(-1, 3).long()
        assert _canon(_triangles_of(lod)) == _canon(li[(li[:, 0] != li[:, 1]) & (li[:, 1] != li[:, 2]) & (li[:, 0] != li[:, 2])])
        for k in range(m.shape[0]):
            v = lod["vidx"][int(m[k, 0]):int(m[k, 0] + m[k, 2])]
            assert v.unique().numel() == v.numel()
    # LOD 0 = the input indices, verbatim (AssetManager_GLTF.cpp:604-606)
    assert torch.equal(lods[0]["indices"].view(-1, 3).long(), tris.long())


def test_lod_chain_follows_the_reference_loop_rules(liboxcull):
    pos, tris = make_mesh("sphere", n=32, seed=2)
    nrm = pos - pos.mean(0)
    nrm = nrm / nrm.norm(dim=1, keepdim=True)
    lods = build_mesh_lods(pos, tris, normals=nrm)
    assert len(lods) >= 4
    assert lods[0]["error"] == 0.0
    for a, b in zip(lods, lods[1:]):
        target = (a["indices"].numel() + 5) // 6 * 3                                         # AssetManager_GLTF.cpp:609
        assert 6 <= b["indices"].numel() <= target + target // 2                             # :639-645
        assert b["error"] >= a["error"] and b["error"] - a["error"] <= 0.5
        assert set(b["indices"].tolist()) <= set(a["indices"].tolist())                     # collapses onto existing vertices only
    # coarser LODs need fewer meshlets
    counts = [l["meshlets"].shape[0] for l in lods]
    assert counts == sorted(counts, reverse=True) and counts[-1] < counts[0]


def test_border_is_locked(liboxcull):
    """meshopt_SimplifyLockBorder: the open boundary of a height field survives every LOD edge for edge."""
    pos, tris = make_mesh("terrain", n=30, seed=3)
    lods = build_mesh_lods(pos, tris)
    assert len(lods) >= 2
    border0 = _border_edges(lods[0]["indices"])
    assert len(border0) > 50
    for lod in lods[1:]:
        assert _border_edges(lod["indices"]) == border0


def test_deterministic_and_limits(liboxcull):
    pos, tris = make_mesh("sphere", n=16, seed=4)
    a = build_mesh_lods(pos, tris)
    b = build_mesh_lods(pos, tris)
    assert len(a) == len(b) and all(torch.equal(x[k], y[k]) for x, y in zip(a, b) for k in ("indices", "meshlets", "vidx", "micro"))
    small = build_mesh_lods(pos, tris, max_lods=2, max_vertices=24, max_triangles=20)
    assert len(small) == 2 and int(small[0]["meshlets"][:, 2].max()) <= 24 and int(small[0]["meshlets"][:, 3].max()) <= 20
    assert small[0]["meshlets"].shape[0] > a[0]["meshlets"].shape[0]


def test_argument_validation(liboxcull):
    pos, tris = make_mesh("sphere", n=8, seed=5)
    bad = tris.clone()
    bad[0, 0] = pos.shape[0]  # index out of range
    with pytest.raises(L.OxcError):
        build_mesh_lods(pos, bad)
    with pytest.raises(L.OxcError):
        build_mesh_lods(pos, tris.reshape(-1)[:-1])  # not a multiple of 3
    with pytest.raises(L.OxcError):
        build_mesh_lods(pos, tris, max_vertices=2)


def _scene_from_mesh(kind, n, instances, seed, bounds_fn, vertex_order=None):
    pos, tris = make_mesh(kind, n=n, seed=seed)
    nrm = pos - pos.mean(0)
    nrm = nrm / nrm.norm(dim=1, keepdim=True).clamp_min(1e-6)
    lods = build_mesh_lods(pos, tris, normals=nrm)
    if vertex_order:
        lods, (pos, nrm) = reorder_vertices(lods, [pos, nrm], by=vertex_order)
    bounds, mesh6, qpos = [], None, None
    for i, lod in enumerate(lods):
        b, m6, q = bounds_fn(pos, lod["meshlets"], lod["vidx"], lod["micro"])
        bounds.append(b)
        if i == 0:
            mesh6, qpos = m6, q  # the mesh AABB and the quantised positions come from LOD 0 (all vertices)
    return lods, make_scene_from_lods(instances, lods, bounds, qpos, mesh6, seed=seed + 100)


@pytest.mark.parametrize("kind,n", [("sphere", 20), ("terrain", 30), ("soup", 25)])
def test_vertex_fetch_remap_moves_ids_not_geometry(liboxcull, kind, n):
    """The asset path's vertex-fetch remap (AssetManager_GLTF.cpp:512-568; here also in meshlet order): a permutation of the vertex ids --
    every triangle of every LOD and every meshlet keeps its corner positions, and in meshlet order a meshlet's position gather touches
    fewer cache lines."""
    pos, tris = make_mesh(kind, n=n, seed=5)
    lods = build_mesh_lods(pos, tris)
    V = pos.shape[0]
    for by in ("meshlets", "indices"):
        remap = vertex_fetch_remap(lods[0]["vidx"] if by == "meshlets" else lods[0]["indices"], V)
        assert sorted(remap.tolist()) == list(range(V))  # a permutation
        used = lods[0]["vidx" if by == "meshlets" else "indices"].reshape(-1).long()
        seen, last = set(), -1
        for v in used.tolist():  # first uses come in increasing new-id order
            if v not in seen:
                seen.add(v)
                assert remap[v] == last + 1
                last += 1
        new_lods, (new_pos,) = reorder_vertices(lods, [pos], by=by)
        for a, b in zip(lods, new_lods):
            assert torch.equal(pos[a["indices"].long()], new_pos[b["indices"].long()])
            assert torch.equal(pos[a["vidx"].long()], new_pos[b["vidx"].long()])
            assert torch.equal(a["meshlets"], b["meshlets"]) and torch.equal(a["micro"], b["micro"])

    def lines_per_gather(ls):  # 128-byte lines of vertex_positions (8 bytes per vertex) one meshlet's vertices live in
        v = ls[0]["vidx"].long()
        return np.mean([len(set((v[o:o + c] * 8 // 128).tolist())) for o, _, c, _ in ls[0]["meshlets"].tolist()])

    if kind != "soup":
        assert lines_per_gather(reorder_vertices(lods, [pos])[0]) < 0.8 * lines_per_gather(lods)


def test_built_chain_through_the_checker_pipeline(liboxcull, oracle_lib):
    """(CPU) the chain is consumable: cull_meshes selects several LODs, the expansion and both cull stages run over them."""
    lods, s = _scene_from_mesh("sphere", 24, 150, 7, oracle.build_meshlet_bounds)
    cam = s.cull_camera()
    mli, _ = oracle.cull_meshes(s, cam, L.CULL_TEST_ALL)
    assert len(set(s.mesh_instances[:, 1].tolist())) > 1 and mli.shape[0] > 0
    vis = oracle.cull_meshlets(s, cam, mli)
    tri = oracle.cull_triangles(s, cam, mli, vis, 0, vis.numel())
    assert vis.numel() > 0 and tri.numel() > 0


@pytest.mark.gpu
def test_built_chain_gpu_producer_and_cull_equal_the_checker(renderer, oracle_lib):
    """triangle soup -> oxc_mesh_build_* (host) -> oxc_build_meshlet_bounds per LOD (GPU) -> cull_meshes (LOD select) -> cull_meshlets ->
    cull_triangles (GPU) == the checker over the same arrays; the GPU-produced bounds == the checker's bounds."""
    from util import assert_same, gpu_frame, oracle_frame

    def gpu_bounds(pos, meshlets, vidx, micro):
        b, m6, q = renderer.build_meshlet_bounds(pos.cuda(), meshlets.cuda(), vidx.cuda(), micro.cuda())
        wb, wm, wq = oracle.build_meshlet_bounds(pos, meshlets, vidx, micro)
        assert torch.equal(b.cpu(), wb) and torch.equal(q.cpu(), wq) and torch.equal(m6.cpu().view(torch.int32), wm.view(torch.int32))
        return b.cpu(), m6.cpu(), q.cpu()

    for kind, n, order in (("sphere", 28, None), ("terrain", 36, None), ("terrain", 36, "meshlets")):
        lods, cpu = _scene_from_mesh(kind, n, 200, 11, gpu_bounds, vertex_order=order)
        gpu = cpu.to("cuda")
        want = oracle_frame(cpu, run_cull_meshes=True)
        got = gpu_frame(renderer, gpu, run_cull_meshes=True)
        assert_same(want, got, ["total", "cull_meshlets_cmd_x", "lod_index", "meshlet_instances", "visible", "indices"])
        assert len(want["visible"]) > 0 and len(want["indices"]) > 0
        if kind == "sphere":
            assert len(set(want["lod_index"].tolist())) > 1
