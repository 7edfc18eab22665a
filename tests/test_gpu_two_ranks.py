"""Two ranks on two GPUs, the exchanges of the path routed through the C ABI's RCCL entry points: rank 0 builds the
pyramid, oxc_broadcast_hiz hands it to rank 1, every rank culls its contiguous shard into shard-local buffers, oxc_pack_counters +
oxc_exchange_counts give every rank all counters.  The union of the shard outputs (ids rebased by the shard's first meshlet)
must be the single-device result of the checker.  Skipped on a one-GPU box (the driver's multi-GPU node runs it)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,top", [(1, 0), (1, 3), (2, 0), (2, 3)], ids=["world1-dry-run", "world1-top-mips", "world2", "world2-top-mips"])
def test_ranks_native_comm_union_equals_single_device(tmp_path, oracle_lib, world, top):
    """world = 1 runs the very same worker on one GPU (RCCL communicator of one rank): keeps the script honest on a one-GPU box."""
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs")
    import oracle
    from oxylus_amd import lib as L
    from oxylus_amd.synth import SceneSpec, make_depth, make_scene

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import two_rank_worker as W
    from util import oracle_hiz

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OXC_TEST_HIZ_TOP=str(top))
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "two_rank_worker.py"), str(r), str(world), str(tmp_path)], env=env) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    ranks = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    # the checker on the whole scene, one device's worth
    full = make_scene(SceneSpec(**W.SPEC), "cpu")
    cam = full.cull_camera()
    hz, levels, offs = oracle_hiz(make_depth(2 * W.HIZ, 2 * W.HIZ, 40, seed=W.DEPTH_SEED), W.HIZ, W.HIZ)
    for r in ranks:
        assert np.array_equal(r["hiz"].view(np.uint32), hz.numpy().view(np.uint32)), "a rank culled against a different pyramid"
    K = W.SPEC["meshlets_per_mesh"]
    N = full.n_meshlet_instances
    v = oracle.Visibility(N, 0, 0)
    out = torch.zeros(N, dtype=torch.int32)
    mask = torch.zeros((N + 31) // 32, dtype=torch.int32)
    hzv = oracle.make_hiz(hz, W.HIZ, W.HIZ, levels, offs)
    for tag, flags in (("early", L.CULL_TEST_ALL), ("late", L.CULL_TEST_ALL | L.CULL_LATE_PASS)):
        n_e = oracle.cull_meshlets_hiz(full, cam, full.meshlet_instances, flags, hzv, v, mask, out)
        first = v.early if tag == "late" else 0
        want_vis = out[first:first + n_e].numpy().astype(np.int64)
        want_idx = oracle.cull_triangles(full, cam, full.meshlet_instances, out, first, n_e).numpy().astype(np.int64) & 0xFFFFFFFF
        got_vis = np.concatenate([r[f"{tag}_visible"].astype(np.int64) + int(r["first_meshlet"]) for r in ranks])
        got_idx = np.concatenate([(r[f"{tag}_indices"].astype(np.int64) & 0xFFFFFFFF) + (int(r["first_meshlet"]) << 8) for r in ranks])
        assert np.array_equal(got_vis, want_vis), tag   # contiguous shards + ascending lists: concatenation IS the global order
        assert np.array_equal(got_idx, want_idx), tag
        # every rank holds every rank's counters, and they are the counters
        for r in ranks:
            assert np.array_equal(r[f"{tag}_gathered"], np.stack([q[f"{tag}_counts"] for q in ranks]))
        assert sum(int(q[f"{tag}_counts"][0]) for q in ranks) == n_e
    assert int(K) > 0
