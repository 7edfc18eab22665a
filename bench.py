#!/usr/bin/env python3
"""bench.py -- meshlets/s culled on MI355X (BASELINE.json metric), one JSON line on rank 0.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one RendererInstance::cull_geometry pass over one batch of synthetic input that is
already resident in HBM.  Default workload = BASELINE.json configs[1]: 1M meshlets (1000 mesh
instances x 1000 meshlets), one reversed-Z perspective camera, frustum + cone cull + ordered
compaction (stages = cull_meshlets only).  The 24 MB working set would sit in the 256 MB Infinity
Cache, so steps rotate over COPIES independent copies of the scene (>= 1 GB) to stay HBM-bound
(SURVEY.md 8d).  `--workload config3` times the full pipeline (HiZ build + two-pass occlusion +
triangle cull) on 10M meshlets instead; it is reported in the same format but is not the
default line.  `--streams S` (default 3 for config 2) keeps S independent batches in flight: S contexts
on S HIP streams inside one HIP graph -- a 1M-meshlet batch is launch/dependency-latency bound, so
consecutive batches are overlapped the way independent views/frames would be; the one-stream figure is
reported next to it as "single_stream".  `--batch B` (default 4) culls B independent frames per
oxc_cull_geometry_batch call: every stage is one launch with grid.y = B (a HIP graph sustains only ~3 us
per kernel node on this platform, tools/launch_rate.py, so launches per frame are what limits a
1M-meshlet batch).  A step is still one frame (one 1M-meshlet batch of synthetic input).

Extra objects on the line: "roofline" (dominant kernel: algorithmic bytes / HIP-event kernel
time vs the 8 TB/s HBM peak) and "cpu_baseline" (the scalar C oracle over the same arrays on
the host cores; a reported baseline, not the target).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from oxylus_amd import lib as L  # noqa: E402
from oxylus_amd.renderer import CullGeometryContext, ImageAttachment, MainGeometryContext, PreparedFrame, RendererInstance  # noqa: E402
from oxylus_amd.synth import SceneSpec, make_depth, make_scene  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=9600)
    ap.add_argument("--warmup", type=int, default=960)
    ap.add_argument("--workload", default="config2", choices=["config2", "config3", "config5", "bounds", "loop", "vsm"])
    ap.add_argument("--tris", type=int, default=64, help="config3: triangles per meshlet; > 64 uses the wide packed index extension (<= 8M meshlets)")
    ap.add_argument("--views", type=int, default=16, help="config5: number of cascade views per step")
    ap.add_argument("--meshlets", type=int, default=0, help="override meshlets per GPU (default 1M / 10M)")
    ap.add_argument("--copies", type=int, default=0, help="independent scene copies rotated through (default: >= 1.1 GB)")
    ap.add_argument("--streams", type=int, default=3, help="independent batches in flight: S contexts on S HIP streams (config2 only)")
    ap.add_argument("--batch", type=int, default=16, help="frames per oxc_cull_geometry_batch call (1 = one call per step; max 16)")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly instead of replaying a HIP graph")
    ap.add_argument("--graph", action="store_true",
                    help="replay a HIP graph even with >= 8 frames per launch (default there: eager launches on real streams, which overlap "
                         "the small prepare/emit kernels of one call with the test kernel of another; measured 1.94e11 vs 1.73e11)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--native-comm", action="store_true",
                    help="multi-GPU: run the two exchanges of the path (counter all-gather, HiZ broadcast) through the C ABI's RCCL entry points "
                         "(oxc_exchange_counts / oxc_broadcast_hiz) instead of torch.distributed; the rendezvous stays torch.distributed")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    return ap.parse_args()


class Step:
    """One pre-marshalled cull_geometry call (C structs built once; the hot loop only calls into
    liboxcull.so)."""

    def __init__(self, r: RendererInstance, scene, stages, use_hiz=False, hiz=None, with_triangles=False, wide=False):
        self.scene = scene
        self.frame = PreparedFrame.create(scene, with_triangles=with_triangles, max_tris=128 if wide else 64)
        self.cframe = self.frame.c()
        self.ctx = CullGeometryContext(use_hiz=use_hiz, init_cull_meshes=False, cull_flags=L.CULL_TEST_ALL,
                                       cull_camera=scene.cull_camera(), hiz_attachment=hiz, stages=stages, wide_triangle_index=wide)
        r.prepared_frame = self.frame
        r.seed_meshlet_instances(self.ctx, scene.n_meshlet_instances)
        self.cctx = self.ctx.c()
        self.pf, self.pc = C.byref(self.cframe), C.byref(self.cctx)


def bench_bounds(args, r, dev, stream, rank, world, dist):
    """--workload bounds: the asset-side meshlet bounds producer (oxc_build_meshlet_bounds) over a
    procedural terrain cut into 8x4-quad patches (64 triangles, 45 vertices per meshlet, vertices not shared
    between patches).  A step = one call over all meshlets of this GPU."""
    import math

    P = args.meshlets or 1_000_000
    steps, warmup = min(args.steps, 50), min(args.warmup, 5)
    with torch.cuda.stream(stream):
        side = int(math.ceil(math.sqrt(P)))
        p = torch.arange(P, device=dev, dtype=torch.int64)
        pi, pj = (p // side).to(torch.float32), (p % side).to(torch.float32)
        v = torch.arange(45, device=dev)
        lu, lv = (v % 9).to(torch.float32), (v // 9).to(torch.float32)
        x = (pj[:, None] * 8 + lu[None, :]) * 0.05
        z = (pi[:, None] * 4 + lv[None, :]) * 0.05
        g = torch.Generator(device=dev).manual_seed(99 + rank)
        y = 0.6 * torch.sin(1.7 * x) * torch.cos(1.3 * z) + 0.02 * torch.randn(x.shape, generator=g, device=dev)
        positions = torch.stack([x, y, z], -1).reshape(-1, 3).contiguous()
        del x, y, z
        corners = []
        for qv in range(4):
            for qu in range(8):
                a, b = qv * 9 + qu, qv * 9 + qu + 1
                d, e = (qv + 1) * 9 + qu, (qv + 1) * 9 + qu + 1
                corners += [a, d, b, b, d, e]
        micro = torch.tensor(corners, dtype=torch.uint8, device=dev).repeat(P).contiguous()
        vidx = torch.arange(45 * P, device=dev, dtype=torch.int32)
        meshlets = torch.stack([p * 45, p * 192, torch.full_like(p, 45), torch.full_like(p, 64)], 1).to(torch.int32).contiguous()
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        for _ in range(warmup):
            out = r.build_meshlet_bounds(positions, meshlets, vidx, micro, stream=stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        for _ in range(steps):
            out = r.build_meshlet_bounds(positions, meshlets, vidx, micro, stream=stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    value = P * world * steps / dt
    # algorithmic bytes per meshlet: Meshlet 16 + 45 vertex ids 180 + 192 micro bytes + 45 float3 540 read,
    # MeshletBounds 16 + {min,max} scratch 24 written and 24 read again by the mesh fold; quantised positions:
    # 540 read + 45 * 8 written
    bytes_per_meshlet = (16 + 180 + 192 + 540 + 16 + 24 + 24) + (540 + 360)
    achieved = bytes_per_meshlet * P * steps / dt / 1e9
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle

        n = min(P, 20_000)
        cp, cm, cv, cmi = positions[: 45 * n].cpu(), meshlets[:n].cpu(), vidx[: 45 * n].cpu(), micro[: 192 * n].cpu()
        tc = time.perf_counter()
        want = oracle.build_meshlet_bounds(cp, cm, cv, cmi)
        t_cal = time.perf_counter() - tc
        reps = int(max(1, min(args.cpu_seconds / max(t_cal, 1e-3), 1000)))
        tc = time.perf_counter()
        for _ in range(reps):
            oracle.build_meshlet_bounds(cp, cm, cv, cmi)
        dtc = time.perf_counter() - tc
        ok = bool(torch.equal(want[0], out[0][:n].cpu()) and torch.equal(want[2], out[2][: 45 * n].cpu()))
        cpu_baseline = {"value": round(n * reps / dtc, 1), "unit": "meshlets/s", "cores": 1, "kind": "port",
                        "sample": f"{reps} passes over the first {n} meshlets of the same arrays, oracle/oxcull_oracle.c orc_build_meshlet_bounds "
                                  f"(sequential), {dtc:.1f} s; GPU records of that range byte-identical: {ok}"}
    if rank == 0:
        print(json.dumps({
            "metric": "meshlets/s bounded (asset-side producer)", "value": round(value, 1), "unit": "meshlets/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 6), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "SURVEY 8f-1: oxc_build_meshlet_bounds over a procedural terrain, 64-triangle / 45-vertex patches",
                       "meshlets_per_gpu": P, "vertices": 45 * P, "quantize_positions": True},
            "roofline": {"bound": "hbm", "kernel": "build_meshlet_bounds (quantize_positions + meshlet_bounds + mesh fold)", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
                         "algorithmic_bytes_per_meshlet": bytes_per_meshlet},
            "cpu_baseline": cpu_baseline}))
    if dist is not None:
        dist.destroy_process_group()


def bench_loop(args, r, dev, stream, rank, world, dist):
    """--workload loop: the closed two-pass frame of RendererInstance::render (RendererInstance.cpp:842-884) without a
    graphics queue -- early cull (last frame's mask) -> oxc_draw_visbuffer -> depth -> oxc_generate_hiz -> late cull ->
    draw on top -- on a static scene (steady state: the early pass draws everything, the late pass finds nothing new)."""
    n_meshlets = args.meshlets or 2_000_000
    K = 1000
    M = max(1, n_meshlets // K)
    n_meshlets = M * K
    W = H = 2048
    steps, warmup = min(args.steps, 30), min(max(args.warmup, 2), 5)
    with torch.cuda.stream(stream):
        scene = make_scene(SceneSpec(n_mesh_instances=M, meshlets_per_mesh=K, with_geometry=True, seed=0x0A1DE5 + 9 + rank), dev)
        r.reserve(M, n_meshlets)
        frame = PreparedFrame.create(scene, with_triangles=True)
        r.prepared_frame = frame
        cam = scene.cull_camera()
        pv = [cam.projection_view[i] for i in range(16)]
        hiz = ImageAttachment.hiz(W // 2, H // 2, dev)
        depth = ImageAttachment.depth(torch.zeros((H, W), dtype=torch.float32, device=dev))
        visdepth = torch.zeros((H, W), dtype=torch.int64, device=dev)
        ctx = CullGeometryContext(use_hiz=True, init_cull_meshes=False, cull_flags=L.CULL_TEST_ALL, cull_camera=cam, hiz_attachment=hiz, stages=L.STAGE_ALL)
        r.seed_meshlet_instances(ctx, n_meshlets)
    from oxylus_amd.renderer import MainGeometryContext

    mg = MainGeometryContext(depth_attachment=depth, hiz_attachment=hiz)
    counts = {}

    def one_frame(record=False):
        ctx.cull_flags = L.CULL_TEST_ALL
        r.cull_geometry(ctx, stream=stream)
        if record:
            c = r.read_counters(ctx, stream=stream)
            counts["early"], counts["early_indices"] = c.cull_triangles_cmd_x, c.draw_index_count
        r.draw_visbuffer(ctx, pv, W, H, visdepth, clear=True, depth=depth, stream=stream)
        r.generate_hiz(mg, stream=stream)
        ctx.cull_flags = L.CULL_TEST_ALL | L.CULL_LATE_PASS
        r.cull_geometry(ctx, stream=stream)
        if record:
            c = r.read_counters(ctx, stream=stream)
            counts["late"], counts["late_indices"] = c.cull_triangles_cmd_x, c.draw_index_count
        r.draw_visbuffer(ctx, pv, W, H, visdepth, clear=False, depth=depth, stream=stream)

    with torch.cuda.stream(stream):
        for _ in range(warmup):
            one_frame()
        one_frame(record=True)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        for _ in range(steps):
            one_frame()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    covered = float((visdepth != 0).float().mean().item())
    if rank == 0:
        print(json.dumps({
            "metric": "meshlets/s through the closed two-pass frame (cull + draw + HiZ)", "value": round(n_meshlets * world * steps / dt, 1), "unit": "meshlets/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 6), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "SURVEY 8f-2 loop: early cull -> draw -> depth -> HiZ -> late cull -> draw, static scene, steady state",
                       "meshlets_per_gpu": n_meshlets, "target": [W, H], "hiz": [W // 2, H // 2], "tris_per_meshlet": 64,
                       "steady_state_counts": counts, "covered_pixel_fraction": round(covered, 4)},
            "roofline": None, "cpu_baseline": None}))
    if dist is not None:
        dist.destroy_process_group()


def bench_vsm(args, r, dev, stream, rank, world, dist):
    """--workload vsm: the virtual-shadow-map cull of draw_virtual_shadowmap (Passes/Shadowmaps.cpp:331-366,433-463):
    oxc_generate_hpb from a page table, then oxc_cull_geometry(use_hpb) = cull_meshes against the coarsest clipmap +
    cull_meshlets_hpb over the 10 dirty clipmap views ("visible if any view's pages want it")."""
    import numpy as np
    from oxylus_amd.renderer import HpbAttachment
    from oxylus_amd.synth import pack_clipmaps, virtual_shadow_matrices

    n_meshlets = args.meshlets or 10_000_000
    K = 1000
    M = max(1, n_meshlets // K)
    n_meshlets = M * K
    steps, warmup = min(args.steps, 50), min(max(args.warmup, 2), 5)
    light = np.array([0.3, -1.0, 0.2])
    light /= np.linalg.norm(light)
    mats, offs, zn = virtual_shadow_matrices([0.0, 0.0, -60.0], light, 500.0, 10.0, 10)
    with torch.cuda.stream(stream):
        scene = make_scene(SceneSpec(n_mesh_instances=M, meshlets_per_mesh=K, with_geometry=False, lod_count=2, seed=0x0A1DE5 + 11 + rank), dev)
        r.reserve(M, n_meshlets)
        frame = PreparedFrame.create(scene, with_triangles=False, expand=False)
        r.prepared_frame = frame
        clip = pack_clipmaps(mats, offs, zn).to(dev)
        g = torch.Generator(device=dev).manual_seed(3 + rank)
        pt = torch.randint(0, 7, (10, 64, 64), generator=g, device=dev, dtype=torch.int32)
        pt[torch.rand((10, 64, 64), generator=g, device=dev) < 0.15] = 7
        hpb = HpbAttachment.create(64, 64, 10, 7, dev)
        dirty = torch.ones(10, dtype=torch.int32, device=dev)
        cam = scene.cull_camera()
        for i in range(16):
            cam.projection_view[i] = float(mats[9][i])
        for i in range(3):
            cam.position[i] = float(-light[i])
        cam.near_clip = zn
        ctx = CullGeometryContext(use_hpb=True, init_cull_meshes=True, cull_flags=L.CULL_TEST_FRUSTUM, cull_camera=cam, hpb_attachment=hpb,
                                  vsm_clipmaps_buffer=clip, vsm_clipmap_dirty_flags_buffer=dirty, vsm_clipmap_count=10,
                                  stages=L.STAGE_MESHES | L.STAGE_MESHLETS)

    def one():
        r.generate_hpb(pt, hpb, stream=stream)
        r.cull_geometry(ctx, stream=stream)

    with torch.cuda.stream(stream):
        for _ in range(warmup):
            one()
    torch.cuda.synchronize()
    c = r.read_counters(ctx, stream)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        for _ in range(steps):
            one()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank == 0:
        print(json.dumps({
            "metric": "meshlets/s culled against 10 clipmap views (VSM page pyramid)", "value": round(n_meshlets * world * steps / dt, 1), "unit": "meshlets/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 6), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "SURVEY 8a-14 + 8f-3: generate_hpb + cull_meshes + cull_meshlets_hpb, 10 dirty clipmaps of 64x64 pages, 15 % pages wanted",
                       "meshlets_per_gpu": n_meshlets, "mesh_instances": M, "after_cull_meshes": c.total_visible_meshlet_instances,
                       "visible": c.cull_triangles_cmd_x},
            "roofline": None, "cpu_baseline": None}))
    if dist is not None:
        dist.destroy_process_group()


def usable_cores() -> int:
    """Host threads this process may actually run at once: the affinity mask, cut down by a cgroup CPU quota when the
    container has one (a box can show 256 CPUs in the mask and still be limited to a few cores' worth of time)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    quota = None
    try:  # cgroup v2: "<quota|max> <period>"
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = int(q) / int(period)
    except (OSError, ValueError):
        try:  # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and period > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        cores = max(1, min(cores, int(quota + 0.999)))
    return cores


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI

    r = RendererInstance(local_rank)
    lib, ctxp = r._lib, r._ctx
    stream = torch.cuda.Stream(device=dev)
    native_comm = bool(args.native_comm and dist is not None)
    if native_comm:
        box = [r.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        r.comm_init(box[0], rank, world)
    if args.workload == "bounds":
        return bench_bounds(args, r, dev, stream, rank, world, dist)
    if args.workload == "loop":
        return bench_loop(args, r, dev, stream, rank, world, dist)
    if args.workload == "vsm":
        return bench_vsm(args, r, dev, stream, rank, world, dist)
    sp = C.c_void_p(stream.cuda_stream)
    n_streams = max(1, args.streams) if args.workload == "config2" else 1
    # extra contexts/streams for independent batches in flight (each context owns its scratch)
    renderers = [r] + [RendererInstance(local_rank) for _ in range(n_streams - 1)]
    streams = [stream] + [torch.cuda.Stream(device=dev) for _ in range(n_streams - 1)]
    sps = [C.c_void_p(s_.cuda_stream) for s_ in streams]

    full = args.workload == "config3"
    multiview = args.workload == "config5"
    n_meshlets = args.meshlets or (10_000_000 if (full or multiview) else 1_000_000)
    wide = full and args.tris > 64
    if wide:
        n_meshlets = min(n_meshlets, 8_000_000)  # 23-bit instance id of the wide index
    K = 1000
    M = max(1, n_meshlets // K)
    n_meshlets = M * K
    spec = SceneSpec(n_mesh_instances=M, meshlets_per_mesh=K, with_geometry=full, seed=0x0A1DE5 + 2 + rank,
                     tris_per_meshlet=(args.tris if full else 64), lod_count=3 if multiview else 1)
    with torch.cuda.stream(stream):
        base = make_scene(spec, dev)
        bytes_per_copy = n_meshlets * 24 + (M * 212)
        if full or multiview:
            copies = args.copies or 1  # 10M meshlets (+ geometry ~10 GB): far beyond the Infinity Cache already
        else:
            copies = args.copies or max(2, -(-1_150_000_000 // bytes_per_copy))
        scenes = [base] + [base.clone() for _ in range(copies - 1)]
        for rr in renderers:
            rr.reserve(M, n_meshlets)
        hiz = depth = None
        if full:
            depth = ImageAttachment.depth(make_depth(8192, 8192, 64, seed=3, device=dev))
            hiz = ImageAttachment.hiz(4096, 4096, dev)
        stages = L.STAGE_ALL if full else (L.STAGE_MESHES | L.STAGE_MESHLETS if multiview else L.STAGE_MESHLETS)
        steps = [Step(renderers[i % n_streams], s, stages, use_hiz=full, hiz=hiz, with_triangles=full, wide=wide) for i, s in enumerate(scenes)]
        if full:
            g = torch.Generator(device=dev).manual_seed(5)
            for st in steps:  # random prior-visibility mask, p = 0.3 (config 3 restatement)
                words = st.frame.meshlet_instance_visibility_mask_buffer.numel()
                bits = (torch.rand((words, 32), generator=g, device=dev) < 0.3).to(torch.int64)
                st.frame.meshlet_instance_visibility_mask_buffer.copy_((bits << torch.arange(32, device=dev)).sum(1).to(torch.int32))
            mask0 = [st.frame.meshlet_instance_visibility_mask_buffer.clone() for st in steps]
    torch.cuda.synchronize()

    def check(st):
        if st != L.OXC_OK:
            raise RuntimeError(lib.oxc_last_error(ctxp).decode())

    if full:
        mg = L.MainGeometryContext()
        mg.struct_size = C.sizeof(L.MainGeometryContext)
        mg.depth_attachment, mg.hiz_attachment = depth.c(), hiz.c()
        pmg = C.byref(mg)

    view_cams = []
    if multiview:
        # config 5: orthographic cascade views with doubling extents around the camera
        # (Shadowmaps.cpp:9-63 generalised to `--views` levels), per-view LOD select (cull_meshes)
        from oxylus_amd.synth import virtual_shadow_matrices

        mats, _, zn = virtual_shadow_matrices([0.0, 0.0, -60.0], [0.3, -1.0, 0.2], 500.0, 2.0, args.views)
        for v in range(args.views):
            cam = base.cull_camera()
            for k in range(16):
                cam.projection_view[k] = float(mats[v][k])
            cam.position[0], cam.position[1], cam.position[2] = 0.0, 0.0, -60.0
            cam.near_clip = zn
            view_cams.append(cam)

    single_stream = [False]  # instrumented pass: every context on stream 0, so kernels do not overlap
    batch = max(1, min(16, args.batch)) if args.workload == "config2" else 1
    groups = []  # (context index k, C arrays) : `batch` copies of the same context culled by ONE batched call
    if batch > 1:
        per_ctx = [[st for i, st in enumerate(steps) if i % n_streams == k] for k in range(n_streams)]
        for k, lst in enumerate(per_ctx):
            for j in range(0, len(lst) - len(lst) % batch, batch):
                grp = lst[j:j + batch]
                cf = (L.PreparedFrame * batch)(*[g_.cframe for g_ in grp])
                cc = (L.CullGeometryContext * batch)(*[g_.cctx for g_ in grp])
                groups.append((k, cf, cc))
        assert groups, "--batch needs at least `batch` copies per stream"
    steps_per_call = batch

    def run_group(gi, n=None):
        k, cf, cc = groups[gi % len(groups)]
        check(lib.oxc_cull_geometry_batch(renderers[k]._ctx, n or batch, cf, cc, sps[0] if single_stream[0] else sps[k]))

    # config 5: the views are independent cull_geometry calls over the same scene; `--batch` of them go through one
    # oxc_cull_geometry_batch call (each element has its own outputs and its own mesh_instances copy: cull_meshes
    # writes lod_index per view)
    view_batch = max(1, min(16, args.batch)) if multiview else 1
    view_groups = []
    if multiview:
        import dataclasses

        with torch.cuda.stream(stream):
            lanes = [steps[0]] + [Step(r, dataclasses.replace(base, mesh_instances=base.mesh_instances.clone()), stages) for _ in range(view_batch - 1)]
            r.reserve(M, n_meshlets)
        for v0 in range(0, args.views, view_batch):
            cams = view_cams[v0:v0 + view_batch]
            cc = (L.CullGeometryContext * len(cams))()
            cf = (L.PreparedFrame * len(cams))()
            for e, cam in enumerate(cams):
                lanes[e].cctx.init_cull_meshes = 1
                lanes[e].cctx.cull_flags = L.CULL_TEST_FRUSTUM | L.CULL_SELECT_LOD
                lanes[e].cctx.cull_camera = cam
                C.memmove(C.byref(cc, e * C.sizeof(L.CullGeometryContext)), lanes[e].pc, C.sizeof(L.CullGeometryContext))
                C.memmove(C.byref(cf, e * C.sizeof(L.PreparedFrame)), lanes[e].pf, C.sizeof(L.PreparedFrame))
            view_groups.append((len(cams), cf, cc))

    def run_step(i):
        st = steps[i % copies]
        if multiview:
            for n, cf, cc in view_groups:
                if n == 1:
                    check(lib.oxc_cull_geometry(ctxp, cf, cc, sp))
                else:
                    check(lib.oxc_cull_geometry_batch(ctxp, n, cf, cc, sp))
            # the counters read below come from view 0's context
            C.memmove(st.pc, C.byref(view_groups[0][2], 0), C.sizeof(L.CullGeometryContext))
            return
        if not full:
            k = (i % copies) % n_streams  # copy -> context/stream
            check(lib.oxc_cull_geometry(renderers[k]._ctx, st.pf, st.pc, sps[0] if single_stream[0] else sps[k]))
            return
        # config 3: HiZ build, then early + late pass against it (render order of
        # RendererInstance.cpp:882-884 restated for a given depth + given mask)
        st.frame.meshlet_instance_visibility_mask_buffer.copy_(mask0[i % copies], non_blocking=True)
        # multi-GPU (configs[3]): depth is produced where rasterisation happens -- rank 0 builds the
        # pyramid and broadcasts it over RCCL/xGMI (89.5 MB); every rank culls its own shard against it
        if world == 1 or rank == 0:
            check(lib.oxc_generate_hiz(ctxp, pmg, sp))
        if dist is not None:
            if native_comm:
                r.broadcast_hiz(hiz, 0, stream)
            else:
                dist.broadcast(hiz.data, src=0)
        st.cctx.cull_flags = L.CULL_TEST_ALL
        check(lib.oxc_cull_geometry(ctxp, st.pf, st.pc, sp))
        st.cctx.cull_flags = L.CULL_TEST_ALL | L.CULL_LATE_PASS
        check(lib.oxc_cull_geometry(ctxp, st.pf, st.pc, sp))

    # ---- parity of what is being timed: copy 0 against the CPU oracle (rank 0) ----
    bit_match = None
    counts = {}
    with torch.cuda.stream(stream):
        run_step(0)
    torch.cuda.synchronize()
    c0 = r.read_counters(steps[0].ctx, stream)
    counts = {"total": c0.total_visible_meshlet_instances, "early": c0.early_visible_meshlet_instances,
              "late": c0.late_visible_meshlet_instances, "emitted": c0.cull_triangles_cmd_x, "index_count": c0.draw_index_count}
    visible_fraction = (c0.cull_triangles_cmd_x if not full else c0.early_visible_meshlet_instances + c0.late_visible_meshlet_instances) / n_meshlets
    units_per_step = n_meshlets * (args.views if multiview else 1)
    cpu_scene = None
    if rank == 0 and not full and not multiview:
        import oracle  # checker only

        oracle.build()
        cpu_scene = base.to("cpu")
        want = oracle.cull_meshlets(cpu_scene, cpu_scene.cull_camera(), cpu_scene.meshlet_instances, nthreads=os.cpu_count() or 1)
        got = steps[0].frame.visible_meshlet_instances_indices_buffer[: c0.cull_triangles_cmd_x].cpu()
        bit_match = bool(want.numel() == got.numel() and torch.equal(want, got))

    # ---- clock ramp (not steps: a fresh box idles at low DPM clocks; the first ~0.5 s of work runs
    # 2-3x slower and would be measured instead of the kernels), then the W warmup steps ----
    ramp = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 0.75:
        with torch.cuda.stream(stream):
            for _ in range(20):
                r.stream_read_probe(ramp, stream)
        torch.cuda.synchronize()
    del ramp
    # a "unit" is one host call: one step, or `batch` steps through oxc_cull_geometry_batch
    def run_unit(u):
        if batch > 1:
            run_group(u)
        else:
            run_step(u)

    units_per_rotation = len(groups) if batch > 1 else copies
    with torch.cuda.stream(stream):
        for u in range(-(-args.warmup // steps_per_call)):
            run_unit(u)
    torch.cuda.synchronize()

    # ---- optional HIP graph over one rotation through the copies ----
    graph = None
    per_replay = units_per_rotation * steps_per_call
    # a HIP graph pays when the loop is launch-bound (few frames per launch); at >= 8 frames per launch there are three
    # launches per ~100 us and the graph executor only gets in the way of cross-stream overlap
    want_graph = not args.no_graph and (args.graph or steps_per_call < 8)
    if want_graph and not full and not multiview and args.steps >= per_replay:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            for s_ in streams[1:]:
                s_.wait_stream(stream)  # fork: the side streams join the capture
            for u in range(units_per_rotation):
                run_unit(u)
            for s_ in streams[1:]:
                stream.wait_stream(s_)  # join
        with torch.cuda.stream(stream):
            for _ in range(max(1, args.warmup // per_replay)):
                graph.replay()
        torch.cuda.synchronize()

    gathered = None
    if dist is not None:
        my_counts = torch.tensor([counts["emitted"], counts["early"], counts["late"], counts["index_count"]], dtype=torch.int32, device=dev)
        gathered = torch.zeros(world * 4, dtype=torch.int32, device=dev)  # flat all-gather target, [world, 4] counters

    def gather_counts():
        if native_comm:
            check(lib.oxc_exchange_counts(ctxp, C.c_void_p(my_counts.data_ptr()), C.c_void_p(gathered.data_ptr()), sp))
        else:
            dist.all_gather_into_tensor(gathered, my_counts)

    def barrier():
        if dist is not None:
            dist.barrier()

    # ---- timed region: EXACTLY args.steps steps ----
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        done = 0
        if graph is not None:
            for _ in range(args.steps // per_replay):
                graph.replay()
                done += per_replay
                if dist is not None:  # per-rank visible counts -> every rank (north star's all-gather), bucketed per rotation
                    gather_counts()
        while args.steps - done >= steps_per_call:
            run_unit(done // steps_per_call)
            done += steps_per_call
            if dist is not None and graph is None and (done // steps_per_call) % units_per_rotation == 0:
                gather_counts()  # same cadence as the graph path: once per rotation through the copies
        if batch > 1 and 1 < args.steps - done:  # remainder smaller than a batch: one shorter batched call
            run_group(done // steps_per_call, args.steps - done)
            done = args.steps
        while done < args.steps:
            run_step(done)
            done += 1
        if dist is not None and graph is None:
            gather_counts()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed_s = float(elapsed.item())
    value = units_per_step * world * args.steps / elapsed_s
    ms_per_step = elapsed_s * 1e3 / args.steps

    # ---- secondary figure: the same steps with ONE batch in flight (one stream, dependent launches) ----
    single = None
    if n_streams > 1 and not full and not multiview and args.steps >= per_replay:
        single_stream[0] = True
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1, stream=stream):
            for u in range(units_per_rotation):
                run_unit(u)
        reps1 = max(2, min(args.steps // per_replay, 20))
        with torch.cuda.stream(stream):
            g1.replay()
        torch.cuda.synchronize()
        t0s = time.perf_counter()
        with torch.cuda.stream(stream):
            for _ in range(reps1):
                g1.replay()
        torch.cuda.synchronize()
        dts = time.perf_counter() - t0s
        single = {"value": round(n_meshlets * reps1 * per_replay / dts, 1), "ms_per_step": round(dts * 1e3 / (reps1 * per_replay), 6),
                  "steps": reps1 * per_replay}
        single_stream[0] = False
        del g1

    # ---- instrumented pass: per-kernel HIP-event times on the same stream, same workload ----
    prof_units = max(1, min(args.steps, max(2 * copies, 96)) // steps_per_call)
    prof_steps = prof_units * steps_per_call
    single_stream[0] = True
    for rr in renderers:
        rr.profile_begin()
    with torch.cuda.stream(stream):
        for u in range(prof_units):
            run_unit(u)
    prof = {"kernels": {}, "empty_pair_ms": 0.0}
    for rr in renderers:
        p_ = rr.profile_end()
        prof["empty_pair_ms"] = max(prof["empty_pair_ms"], p_["empty_pair_ms"])
        for name, k in p_["kernels"].items():
            acc = prof["kernels"].setdefault(name, {"launches": 0, "total_ms": 0.0})
            acc["launches"] += k["launches"]
            acc["total_ms"] += k["total_ms"]
    single_stream[0] = False
    kernels = {}
    for name, k in prof["kernels"].items():
        # raw event-to-event time per launch (what rocprofv3's kernel duration also spans: dispatch + run);
        # the cost of an empty event pair is reported separately, not subtracted
        avg_us = (k["total_ms"] / k["launches"]) * 1e3
        kernels[name] = {"launches_per_step": k["launches"] / prof_steps, "avg_us": round(avg_us, 3)}
    kernels["_empty_event_pair_us"] = round(prof["empty_pair_ms"] * 1e3, 3)

    # measured streaming-read ceiling of this GPU (16 B/lane sum kernel over 2 GiB)
    probe = torch.empty(2 << 30, dtype=torch.uint8, device=dev)
    probe.random_(0, 255)
    with torch.cuda.stream(stream):
        t_probe = time.perf_counter()  # the memory clock needs sustained streaming before the rate settles (a short --steps run
        while time.perf_counter() - t_probe < 0.3:  # leaves the GPU idling at low clocks by the time it gets here: 2.0 instead of 6.0 TB/s)
            for _ in range(30):
                r.stream_read_probe(probe, stream)
            stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(30):
            r.stream_read_probe(probe, stream)
        e1.record(stream)
    torch.cuda.synchronize()
    stream_read_gbps = 30 * probe.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del probe

    # ---- roofline of the dominant kernel ----
    if not full:
        dom = "cull_meshlets_test"  # (config5: one launch per view; bytes are per launch over the LOD-selected list)
        # 8 B MeshletInstance + 16 B MeshletBounds read per meshlet + per-mesh tables
        # 212/K B + 4*v B of visible indices written (the write is done by cull_meshlets_emit; it is
        # charged to the stage, i.e. to this launch, as 8(d) does).
        bytes_per_unit = 24.0 + 212.0 / K + 4.0 * visible_fraction
        units = n_meshlets * steps_per_call  # one launch covers `batch` frames
    else:
        dom = "cull_triangles_test"
        v_tot = counts["early"] + counts["late"]
        bytes_per_unit = 4 + 8 + 16 + (3 * args.tris + 3) // 4 * 4 + 4 * 64 + 8 * 64  # 988 B per visible meshlet (V=64, T=64; 1168 B at T=124), SURVEY 8(d) a11
        units = v_tot / 2.0  # two launches (early, late) share the visible set
    dom_us = (kernels.get(dom) or {}).get("avg_us")
    roofline = None
    if dom_us and dom_us > 0:
        achieved = bytes_per_unit * units / (dom_us * 1e-6) / 1e9
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None, "traffic_source": None,
                    "algorithmic_bytes_per_launch": round(bytes_per_unit * units), "kernel_avg_us": dom_us,
                    "measured_stream_read_GBps": round(stream_read_gbps, 1),  # plain 16 B/lane loads; `nt` loads stream at 6.8-7.1 TB/s
                    "frac_of_measured_stream_read": round(achieved / stream_read_gbps, 4)}  # (profiles/r01_bw_probe.txt)

    # HBM traffic of the dominant kernel from the committed rocprofv3 --pmc summary (collected in its own
    # passes, FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024 as MI355X_MICROARCH.md prescribes); PMC counters
    # cannot be read from inside this process.
    if roofline is not None:
        try:
            prof_name = {"config2": "r01_config2_pmc.json", "config3": "r01_config3_pmc.json", "config5": "r01_config5_pmc.json"}[args.workload]
            with open(os.path.join(ROOT, "profiles", prof_name)) as fpm:
                pm = json.load(fpm)
            per_variant = []  # config3 launches the early and the late instantiation once each per step:
            for kname, cs in pm.get("pmc", {}).items():  # kernel_avg_us averages both, so does traffic
                ok_variant = args.workload != "config2" or ("_batch" in kname) == (steps_per_call > 1) and ("_batch" in kname or "<false, false, false" in kname)
                if dom in kname and ok_variant and "hbm_read_bytes_corrected" in cs:
                    per_variant.append(cs["hbm_read_bytes_corrected"] + cs.get("hbm_write_bytes", 0))
                    if args.workload == "config2":
                        break
            if per_variant:
                roofline["traffic"] = round(sum(per_variant) / len(per_variant))
                roofline["traffic_source"] = f"profiles/{prof_name} (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, separate passes, per launch)"
        except (OSError, KeyError, ValueError):
            pass

    # ---- CPU baseline: the scalar C oracle over the same arrays, all host cores ----
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and cpu_scene is not None:
        import oracle

        cores = usable_cores()
        cam = cpu_scene.cull_camera()
        t1c0 = time.perf_counter()
        oracle.cull_meshlets(cpu_scene, cam, cpu_scene.meshlet_instances, nthreads=1)
        dt1 = time.perf_counter() - t1c0
        # calibrate on a short multi-threaded run (thread scaling on the box is not known in advance),
        # then size the sample to ~cpu_seconds of wall time
        cal = 8
        while True:  # grow the calibration run until thread start-up no longer dominates it
            tc = time.perf_counter()
            oracle.cull_meshlets(cpu_scene, cam, cpu_scene.meshlet_instances, nthreads=cores, passes=cal)
            t_cal = time.perf_counter() - tc
            if t_cal > 0.5 or cal >= 4096:
                break
            cal *= 4
        passes = int(min(max(1, args.cpu_seconds / (t_cal / cal)), 1_000_000))
        t_cpu0 = time.perf_counter()
        oracle.cull_meshlets(cpu_scene, cam, cpu_scene.meshlet_instances, nthreads=cores, passes=passes)
        dt = time.perf_counter() - t_cpu0
        cpu_baseline = {"value": round(n_meshlets * passes / dt, 1), "unit": "meshlets/s", "cores": cores, "kind": "port",
                        "sample": f"{passes} passes over the same {n_meshlets}-meshlet scene (copy 0), oracle/oxcull_oracle.c "
                                  f"orc_cull_meshlets_mt_passes, static range split over {cores} pthreads, {dt:.1f} s",
                        "single_thread_value": round(n_meshlets / dt1, 1)}

    if rank == 0:
        line = {
            "metric": "meshlets/s culled",
            "value": round(value, 1),
            "unit": "meshlets/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 6),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": ("configs[1]: 1M meshlets, one camera, frustum+cone cull + ordered compaction (cull_meshlets stage)"
                             if not (full or multiview) else
                             "configs[2]: 10M meshlets + 4096^2 HiZ (13 mips) from 8192^2 depth: hiz build + early/late occlusion cull + triangle cull + compaction"
                             if full else
                             f"configs[4]: 10M meshlets x {args.views} orthographic cascade views, per-view cull_meshes (frustum + LOD select) + cull_meshlets"),
                "meshlets_per_gpu": n_meshlets, "mesh_instances": M, "meshlets_per_mesh": K, "tris_per_meshlet": args.tris if full else None,
                "copies_rotated": copies, "working_set_MB": round(copies * bytes_per_copy / 1e6, 1),
                "hip_graph": graph is not None, "streams": n_streams, "frames_per_launch": (view_batch if multiview else steps_per_call), "visible_fraction": round(visible_fraction, 4),
                "sharding": (f"contiguous range per rank x{world}; all-gather of per-rank counters"
                             + ("; HiZ built on rank 0 and broadcast" if full else "")) if world > 1 else "single GPU",
            },
            "bit_match": bit_match,
            "single_stream": single,
            "counts": counts,
            "kernels": kernels,
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()
    for rr in renderers:
        rr.close()


if __name__ == "__main__":
    main()
