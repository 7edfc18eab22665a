"""Worker of tests/test_gpu_two_ranks.py: one process per GPU, the multi-GPU exchanges of SURVEY 8e through the C ABI only
(oxc_comm_unique_id / oxc_comm_init / oxc_broadcast_hiz / oxc_pack_counters / oxc_exchange_counts -- RCCL, no torch.distributed).

  python tests/two_rank_worker.py <rank> <world> <rendezvous dir>
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oxylus_amd import lib as L  # noqa: E402
from oxylus_amd.renderer import CullGeometryContext, ImageAttachment, MainGeometryContext, PreparedFrame, RendererInstance  # noqa: E402
from oxylus_amd.shard import shard_scene  # noqa: E402
from oxylus_amd.synth import SceneSpec, make_depth, make_scene  # noqa: E402

SPEC = dict(n_mesh_instances=96, meshlets_per_mesh=250, seed=0x0A1DE5 + 3)
HIZ, DEPTH_SEED = 512, 13


def main():
    rank, world, rdv = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    torch.cuda.set_device(rank)
    dev = f"cuda:{rank}"
    full = make_scene(SceneSpec(**SPEC), "cpu")
    shard_cpu, first_meshlet = shard_scene(full, rank, world)
    shard = shard_cpu.to(dev)
    r = RendererInstance(rank)
    uid_path = os.path.join(rdv, "uid.bin")
    if rank == 0:
        uid = r.comm_unique_id()
        with open(uid_path + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(uid_path + ".tmp", uid_path)
    else:
        t0 = time.time()
        while not os.path.exists(uid_path):
            if time.time() - t0 > 120:
                raise SystemExit("rank 0 never published the RCCL unique id")
            time.sleep(0.05)
        uid = open(uid_path, "rb").read()
    r.comm_init(uid, rank, world)
    hiz = ImageAttachment.hiz(HIZ, HIZ, dev)
    top = int(os.environ.get("OXC_TEST_HIZ_TOP", "0"))  # > 0: the "top mips" exchange (levels >= top travel, the rest is built locally)
    if rank == 0 or top:  # depth is produced where rasterisation happens: rank 0 builds the pyramid ...
        depth = make_depth(2 * HIZ, 2 * HIZ, 40, seed=DEPTH_SEED, device=dev)
        if rank == 0:
            r.generate_hiz(MainGeometryContext(ImageAttachment.depth(depth), hiz))
        else:  # ... or, in the top-mips form, a rank builds the levels below `top` from its own copy of the depth image
            low = ImageAttachment(hiz.data, hiz.width, hiz.height, top, hiz.level_offset[:top])
            r.generate_hiz(MainGeometryContext(ImageAttachment.depth(depth), low))
    r.broadcast_hiz(hiz, 0, first_level=top)  # ... and every rank receives it
    frame = PreparedFrame.create(shard)
    r.prepared_frame = frame
    ctx = CullGeometryContext(use_hiz=True, init_cull_meshes=False, cull_flags=L.CULL_TEST_ALL, cull_camera=shard.cull_camera(), hiz_attachment=hiz)
    r.seed_meshlet_instances(ctx, shard.n_meshlet_instances)
    out = {"first_meshlet": first_meshlet, "hiz": hiz.data.cpu().numpy()}
    mine = torch.zeros(4, dtype=torch.int32, device=dev)
    for tag, flags in (("early", L.CULL_TEST_ALL), ("late", L.CULL_TEST_ALL | L.CULL_LATE_PASS)):
        ctx.cull_flags = flags
        r.cull_geometry(ctx)
        r.pack_counters(ctx, mine)
        gathered = r.exchange_counts(mine)  # all-gather of {emitted, early, late, index_count} over RCCL
        c = r.read_counters(ctx)
        first = c.early_visible_meshlet_instances if tag == "late" else 0
        out[f"{tag}_visible"] = frame.visible_meshlet_instances_indices_buffer[first:first + c.cull_triangles_cmd_x].cpu().numpy()
        out[f"{tag}_indices"] = frame.reordered_indices_buffer[:c.draw_index_count].cpu().numpy()
        out[f"{tag}_counts"] = np.array([c.cull_triangles_cmd_x, c.early_visible_meshlet_instances, c.late_visible_meshlet_instances, c.draw_index_count], dtype=np.int32)
        out[f"{tag}_gathered"] = gathered.cpu().numpy()
    np.savez(os.path.join(rdv, f"rank{rank}.npz"), **out)
    r.comm_destroy()
    r.close()


if __name__ == "__main__":
    main()
