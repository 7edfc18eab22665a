"""Multi-GPU sharding of the cull path: contiguous ranges of mesh instances, one
process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in
the CPU tests).  No data-path collective: every rank culls its own shard into shard-local
compacted buffers with shard-local 24-bit ids.  The two exchanges are an all-gather of the
per-rank counters and a broadcast of the HiZ pyramid from the rank that built it.
"""
from __future__ import annotations

from typing import List, Tuple

import torch

from .synth import Scene


def shard_ranges(n_mesh_instances: int, world: int, block: int = 0):
    """block == 0: contiguous [begin, end) mesh-instance ranges, one per rank, split on instance boundaries so that the
    meshlet_instance_visibility_offset ranges of different ranks stay disjoint -- List[(begin, end)].
    block > 0 ("interleaved range assignment in blocks of e.g. 64k meshlets if contiguous split is too skewed"): the
    instances are cut into blocks of `block` and dealt round robin -- rank r owns blocks r, r + world, ...; List[List[(begin, end)]],
    a rank's ranges in ascending order (possibly empty).  Visibility is spatially coherent, so a contiguous eighth of a scene can hold
    most of what the camera sees; blocks of 64 instances x 1 000 meshlets spread that over the ranks."""
    if block <= 0:
        return [(n_mesh_instances * r // world, n_mesh_instances * (r + 1) // world) for r in range(world)]
    out = [[] for _ in range(world)]
    for k, a in enumerate(range(0, n_mesh_instances, block)):
        out[k % world].append((a, min(a + block, n_mesh_instances)))
    return out


def shard_id_map(scene: Scene, ranges) -> torch.Tensor:
    """For a shard made of several instance ranges (Scene.take): int64 [pieces, 2] = {first shard-local meshlet-instance index of the
    piece, global index of its first meshlet instance}; local id i of piece p is global id i - map[p, 0] + map[p, 1]."""
    col = scene.meshlet_instances[:, 0]
    rows, local = [], 0
    for a, b in ranges:
        lo = int((col < a).sum().item())
        hi = int((col < b).sum().item())
        rows.append((local, lo))
        local += hi - lo
    return torch.tensor(rows, dtype=torch.int64).view(-1, 2)


def shard_scene(scene: Scene, rank: int, world: int, block: int = 0):
    """The sub-scene rank `rank` owns and what maps its shard-local meshlet-instance ids back to global ones: the global index of its
    first meshlet instance (contiguous split) or the piece table of shard_id_map (block > 0: interleaved blocks; an empty shard gives
    (None, empty table)).  Shard-local ids: mesh_instance_index and visibility offsets are rebased to the shard."""
    if block > 0:
        assert scene.spec.share_meshes == 0, "interleaved blocks need one mesh per instance (Scene.take)"
        mine = shard_ranges(scene.n_mesh_instances, world, block)[rank]
        if not mine:
            return None, torch.zeros((0, 2), dtype=torch.int64)
        return scene.take(mine), shard_id_map(scene, mine)
    a, b = shard_ranges(scene.n_mesh_instances, world)[rank]
    K = scene.spec.meshlets_per_mesh
    if scene.spec.share_meshes == 0:  # one mesh per instance: the rank holds ONLY its range of every array
        first_meshlet = int((scene.meshlet_instances[:, 0] < a).sum().item())
        return scene.slice(a, b), first_meshlet
    # shared meshes: instances of any range may reference any mesh -- the (small) mesh table and its geometry stay whole
    kw = {}
    for name in ("bounds", "meshlets", "micro", "vidx", "positions", "lods", "meshes", "transforms"):
        kw[name] = getattr(scene, name).clone()
    mesh_instances = scene.mesh_instances[a:b].clone()
    first_meshlet = a * K
    mesh_instances[:, 4] -= first_meshlet  # rebase; stays a multiple of K, ranks stay disjoint
    mli = scene.meshlet_instances[a * K:b * K].clone()
    mli[:, 0] -= a
    s = Scene(spec=scene.spec, device=scene.device, camera=scene.camera, n_meshes=scene.n_meshes,
              lod_meshlet_counts=scene.lod_meshlet_counts, _lod_tables=scene._lod_tables,
              mesh_instances=mesh_instances.contiguous(), meshlet_instances=mli.contiguous(), **kw)
    return s.bind(), first_meshlet


def exchange_counts(local_counts: torch.Tensor, group=None) -> torch.Tensor:
    """All-gather of the per-rank counters {emitted, early, late, index_count} (int32[4]).
    Returns int32 [world, 4]; every rank derives its exclusive prefix from it."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    n = local_counts.numel()
    out = torch.zeros(world * n, dtype=local_counts.dtype, device=local_counts.device)  # flat: gloo insists on it
    dist.all_gather_into_tensor(out, local_counts.contiguous().view(-1), group=group)
    return out.view(world, n)


def exclusive_offsets(all_counts: torch.Tensor) -> torch.Tensor:
    """Exclusive prefix over ranks: where each rank's lists start in a merged global list."""
    return torch.cumsum(all_counts, 0) - all_counts


def broadcast_hiz(hiz_data: torch.Tensor, src: int = 0, group=None) -> None:
    """Broadcast the whole pyramid (89.5 MB for 4096^2) from the rank that built it.  The
    "top mips only" variant is not used: a rank must never sample a mip it does not hold and
    clamping would change results."""
    import torch.distributed as dist

    dist.broadcast(hiz_data, src=src, group=group)


def broadcast_hiz_top(hiz_data: torch.Tensor, level_offset_bytes, first_level: int, src: int = 0, group=None) -> None:
    """The "top mips" form (north star wording): only levels >= first_level travel; a rank that uses it must have built the
    levels below from its own copy of the depth image (same bytes: the pyramid is a pure function of the depth)."""
    import torch.distributed as dist

    dist.broadcast(hiz_data[level_offset_bytes[first_level] // 4:], src=src, group=group)


def merge_visible(local_visible: torch.Tensor, first_meshlet) -> torch.Tensor:
    """Shard-local visible ids -> global meshlet-instance ids.  first_meshlet: the shard's first global id (contiguous split) or the
    piece table of shard_id_map (interleaved blocks: each id moves by its own piece's offset)."""
    ids = local_visible.to(torch.int64)
    if isinstance(first_meshlet, torch.Tensor):
        if first_meshlet.numel() == 0:
            return ids
        piece = torch.searchsorted(first_meshlet[:, 0].contiguous(), ids, right=True) - 1
        return ids - first_meshlet[piece, 0] + first_meshlet[piece, 1]
    return ids + first_meshlet
