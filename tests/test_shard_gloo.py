"""Multi-GPU path on CPU: world_size 2, gloo.  Each rank culls its contiguous shard (with the
CPU oracle standing in for the device), ranks all-gather their counters and the union of the
shard outputs (with rank offsets) must equal the single-process result (Appendix B.10)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, top=0, block=0):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from oxylus_amd.shard import broadcast_hiz, broadcast_hiz_top, exchange_counts, exclusive_offsets, merge_visible, shard_scene
    from oxylus_amd.synth import SceneSpec, make_depth, make_scene
    from util import oracle_frame, oracle_hiz

    spec = SceneSpec(n_mesh_instances=21, meshlets_per_mesh=96, seed=77)
    scene = make_scene(spec, "cpu")  # every rank generates the same scene, keeps only its shard
    shard, first = shard_scene(scene, rank, world, block)  # (block > 0: interleaved blocks of instances, `first` is then the piece table)
    # HiZ: rank 0 builds, everyone receives
    levels, offs, total = __import__("oxylus_amd.synth", fromlist=["hiz_layout"]).hiz_layout(128, 128)
    hz = torch.zeros(total // 4)
    if rank == 0:
        hz, _, _ = oracle_hiz(make_depth(256, 256, 32, seed=5), 128, 128)
    elif top:  # "top mips" exchange: this rank builds the levels below `top` from its own copy of the depth image
        oracle.generate_hiz(make_depth(256, 256, 32, seed=5), hz, 128, 128, top, offs[:top])
    if top:
        broadcast_hiz_top(hz, offs, top, src=0)
    else:
        broadcast_hiz(hz, src=0)
    hizd = {"data": hz, "w": 128, "h": 128, "levels": levels, "offs": offs}
    mask = torch.zeros((shard.n_meshlet_instances + 31) // 32, dtype=torch.int32)
    res = oracle_frame(shard, use_hiz=True, hiz=hizd, mask=mask, two_pass=True)
    local = torch.tensor([res["late_emitted"], res["early"], res["late"], len(res["late_indices"])], dtype=torch.int32)
    allc = exchange_counts(local)
    offs_r = exclusive_offsets(allc)
    glob = merge_visible(torch.from_numpy(res["late_visible"]), first)
    # packed triangle indices carry shard-local 24-bit ids; globalise them the same way
    idx = torch.from_numpy(res["late_indices"]).to(torch.int64)
    idx_glob = (merge_visible(idx >> 8, first) << 8) | (idx & 0xFF)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), counts=allc.numpy(), offsets=offs_r.numpy(), visible=glob.numpy(), indices=idx_glob.numpy(),
             hiz_sum=float(hz.double().sum()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("block", [0, 4], ids=["contiguous", "interleaved-blocks-of-4-instances"])
@pytest.mark.parametrize("top", [0, 2], ids=["whole-pyramid", "top-mips"])
def test_two_rank_shards_union_equals_single(tmp_path, oracle_lib, top, block):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), top, block), nprocs=world, join=True)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oxylus_amd.synth import SceneSpec, make_depth, make_scene
    from util import oracle_frame, oracle_hiz

    scene = make_scene(SceneSpec(n_mesh_instances=21, meshlets_per_mesh=96, seed=77), "cpu")
    hz, levels, offs = oracle_hiz(make_depth(256, 256, 32, seed=5), 128, 128)
    mask = torch.zeros((scene.n_meshlet_instances + 31) // 32, dtype=torch.int32)
    single = oracle_frame(scene, use_hiz=True, hiz={"data": hz, "w": 128, "h": 128, "levels": levels, "offs": offs}, mask=mask, two_pass=True)
    r = [np.load(os.path.join(str(tmp_path), f"rank{k}.npz")) for k in range(world)]
    # every rank sees the same gathered counters, and they sum to the single-process totals
    assert np.array_equal(r[0]["counts"], r[1]["counts"])
    assert r[0]["counts"][:, 0].sum() == single["late_emitted"]
    assert r[0]["counts"][:, 3].sum() == len(single["late_indices"])
    assert r[0]["offsets"][1, 0] == r[0]["counts"][0, 0]
    vis = np.concatenate([r[0]["visible"], r[1]["visible"]])
    idx = np.concatenate([r[0]["indices"], r[1]["indices"]])
    if block:  # interleaved blocks: every rank's list ascends, the global list is their merge
        assert all(np.all(np.diff(r[k]["visible"]) > 0) for k in range(world))
        vis, idx = np.sort(vis), np.sort(idx)
    # contiguous ranges + ascending shard-local order => concatenation IS the global ascending list
    assert np.array_equal(vis, single["late_visible"].astype(np.int64))
    assert np.array_equal(idx, single["late_indices"].astype(np.int64))
    # the broadcast delivered rank 0's pyramid
    assert r[0]["hiz_sum"] == r[1]["hiz_sum"] == float(hz.double().sum())


def _worker_views(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from oxylus_amd import lib as L
    from oxylus_amd.shard import exchange_counts, exclusive_offsets, shard_scene
    from oxylus_amd.synth import SceneSpec, make_scene, virtual_shadow_matrices

    scene = make_scene(SceneSpec(n_mesh_instances=23, meshlets_per_mesh=80, lod_count=2, seed=78, with_geometry=False), "cpu")
    shard, _ = shard_scene(scene, rank, world)
    mats, _, zn = virtual_shadow_matrices([0.0, 0.0, -60.0], [0.3, -1.0, 0.2], 500.0, 2.0, 4)
    flags = L.CULL_TEST_FRUSTUM | L.CULL_SELECT_LOD
    local, lists = [], []
    for v in range(4):  # configs[4]: every view runs cull_meshes (frustum + LOD select) + cull_meshlets over the rank's range
        s = shard.clone()
        cam = s.cull_camera()
        for k in range(16):
            cam.projection_view[k] = float(mats[v][k])
        cam.position[0], cam.position[1], cam.position[2] = 0.0, 0.0, -60.0
        cam.near_clip = zn
        mli, _ = oracle.cull_meshes(s, cam, flags)
        vis = oracle.cull_meshlets(s, cam, mli)
        local += [vis.numel(), mli.shape[0]]
        lists.append(vis.numpy())
    allc = exchange_counts(torch.tensor(local, dtype=torch.int32))  # [world, views * 2]: {visible, processed} per view
    offs = exclusive_offsets(allc)
    # a view's global MeshletInstance list is the concatenation of the ranks' lists (contiguous instance ranges): local id + the lists before it
    np.savez(os.path.join(out_dir, f"views{rank}.npz"), counts=allc.numpy(), **{f"v{v}": lists[v].astype(np.int64) + int(offs[rank, 2 * v + 1]) for v in range(4)})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_multiview_shards_union_equals_single(tmp_path, oracle_lib):
    """configs[4] sharded: contiguous instance ranges per rank, per-view {visible, processed} counts all-gathered,
    union of the per-view visible lists (ids rebased by the gathered list lengths) == the single-process result of every view."""
    import oracle
    from oxylus_amd import lib as L
    from oxylus_amd.synth import SceneSpec, make_scene, virtual_shadow_matrices

    world = 2
    mp.spawn(_worker_views, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    scene = make_scene(SceneSpec(n_mesh_instances=23, meshlets_per_mesh=80, lod_count=2, seed=78, with_geometry=False), "cpu")
    mats, _, zn = virtual_shadow_matrices([0.0, 0.0, -60.0], [0.3, -1.0, 0.2], 500.0, 2.0, 4)
    r = [np.load(os.path.join(str(tmp_path), f"views{k}.npz")) for k in range(world)]
    assert np.array_equal(r[0]["counts"], r[1]["counts"])
    seen = 0
    for v in range(4):
        s = scene.clone()
        cam = s.cull_camera()
        for k in range(16):
            cam.projection_view[k] = float(mats[v][k])
        cam.position[0], cam.position[1], cam.position[2] = 0.0, 0.0, -60.0
        cam.near_clip = zn
        mli, _ = oracle.cull_meshes(s, cam, L.CULL_TEST_FRUSTUM | L.CULL_SELECT_LOD)
        vis = oracle.cull_meshlets(s, cam, mli)
        assert int(r[0]["counts"][:, 2 * v].sum()) == vis.numel() and int(r[0]["counts"][:, 2 * v + 1].sum()) == mli.shape[0], f"view {v}: gathered counts"
        assert np.array_equal(np.concatenate([r[0][f"v{v}"], r[1][f"v{v}"]]), vis.numpy().astype(np.int64)), f"view {v}: union of the shard lists"
        seen += vis.numel()
    assert seen > 30  # (the small cascades see little of a 23-instance scene; the wide ones most of it)


@pytest.mark.parametrize("block", [0, 8], ids=["contiguous", "blocks-of-8"])
def test_one_scene_shards_are_placed_on_the_whole_scenes_grid(block):
    """bench.py --one-scene: a rank generates only its own instances but PLACES them where the whole scene's grid puts their global indices
    (make_scene(global_ids, global_total)): every instance of every shard sits in the grid cell of its global index, the cells of all ranks
    together are all cells, and contiguous ranges are depth slabs (the generator's grid runs far to near with the index)."""
    import math

    from oxylus_amd.shard import shard_ranges
    from oxylus_amd.synth import SceneSpec, make_scene

    M, world, D = 216, 4, 200.0
    n_side = math.ceil(M ** (1.0 / 3.0))
    seen, mean_z = [], []
    for rank in range(world):
        mine = shard_ranges(M, world, block)[rank]
        mine = [mine] if block <= 0 else mine
        gids = torch.cat([torch.arange(a, b, dtype=torch.int64) for a, b in mine])
        sc = make_scene(SceneSpec(n_mesh_instances=int(gids.numel()), meshlets_per_mesh=4, with_geometry=False, seed=11 + rank, scene_depth=D), "cpu",
                        global_ids=gids, global_total=M)
        pos = sc.transforms.view(-1, 4, 4)[:, 3, 0:3]
        u = torch.stack([pos[:, 0] / D + 0.5, pos[:, 1] / D + 0.5, (pos[:, 2] + D) / (1.1 * D)], 1) * n_side
        cell = torch.floor(u).to(torch.int64).clamp_(0, n_side - 1)
        want = torch.stack([gids % n_side, (gids // n_side) % n_side, gids // (n_side * n_side)], 1)
        assert torch.equal(cell, want), f"rank {rank}"
        seen.append(gids)
        mean_z.append(float(pos[:, 2].mean()))
    assert torch.equal(torch.sort(torch.cat(seen))[0], torch.arange(M))
    if block == 0:
        assert all(mean_z[k] < mean_z[k + 1] for k in range(world - 1))  # slabs, far to near
    else:
        assert max(mean_z) - min(mean_z) < 0.25 * D                        # interleaved blocks: every rank spans the depth range


def test_interleaved_shard_ranges_cover_and_disjoint():
    from oxylus_amd.shard import shard_ranges

    for n, w, blk in ((21, 2, 4), (100, 8, 7), (5, 8, 1), (64, 4, 64), (1000, 8, 64)):
        per_rank = shard_ranges(n, w, blk)
        assert len(per_rank) == w
        flat = sorted(x for rs in per_rank for x in rs)
        assert flat[0][0] == 0 and flat[-1][1] == n and all(flat[i][1] == flat[i + 1][0] for i in range(len(flat) - 1))
        assert all(b - a <= blk for a, b in flat)
        sizes = [sum(b - a for a, b in rs) for rs in per_rank]
        assert max(sizes) - min(sizes) <= blk


def test_shard_ranges_cover_and_disjoint():
    from oxylus_amd.shard import shard_ranges

    for n in (1, 7, 8, 1000, 12345):
        for w in (1, 2, 4, 8):
            rs = shard_ranges(n, w)
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
