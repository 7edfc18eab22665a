"""Auxiliary bench workloads (`bench.py --workload bounds|loop|vsm|config5`): the rows of SURVEY 8f around the hot path and
BASELINE configs[4].  Same JSON shape as the main line; none of them is the driver's default."""
import ctypes as C
import json
import time

import torch

from oxylus_amd import lib as L
from oxylus_amd.renderer import CullGeometryContext, ImageAttachment, MainGeometryContext, PreparedFrame
from oxylus_amd.synth import SceneSpec, make_scene

HBM_PEAK_GBPS = 8000.0


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
    r.profile_begin()
    with torch.cuda.stream(stream):
        for _ in range(10):
            one_frame()
    prof = r.profile_end()
    per_frame_us = {k: round(v["total_ms"] / 10 * 1e3, 1) for k, v in prof["kernels"].items() if not k.startswith("_")}
    if rank == 0:
        print(json.dumps({
            "metric": "meshlets/s through the closed two-pass frame (cull + draw + HiZ)", "value": round(n_meshlets * world * steps / dt, 1), "unit": "meshlets/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 6), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "SURVEY 8f-2 loop: early cull -> draw -> depth -> HiZ -> late cull -> draw, static scene, steady state",
                       "meshlets_per_gpu": n_meshlets, "target": [W, H], "hiz": [W // 2, H // 2], "tris_per_meshlet": 64,
                       "steady_state_counts": counts, "covered_pixel_fraction": round(covered, 4), "per_frame_us": per_frame_us},
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




def bench_config5(args, r, dev, stream, rank, world, dist):
    """--workload config5 = BASELINE configs[4]: 10M meshlets x `--views` orthographic cascade views (Shadowmaps.cpp:9-63
    generalised: doubling extents around the camera), per-view cull_meshes (frustum + LOD select, cull_meshes.slang:35-57) +
    cull_meshlets, up to 16 views per oxc_cull_geometry_batch call.  `value` counts the meshlets the meshlet stage actually
    PROCESSED (per view: the list cull_meshes produced), not candidates x views."""
    import dataclasses

    from oxylus_amd.synth import virtual_shadow_matrices

    n_meshlets = args.meshlets or 10_000_000
    K = 1000
    M = max(1, n_meshlets // K)
    n_meshlets = M * K
    views, vb = args.views, max(1, min(16, args.batch))
    steps, warmup = min(args.steps, 200), min(max(args.warmup, 2), 10)
    lib, ctxp, sp = r._lib, r._ctx, C.c_void_p(stream.cuda_stream)
    with torch.cuda.stream(stream):
        base = make_scene(SceneSpec(n_mesh_instances=M, meshlets_per_mesh=K, with_geometry=False, lod_count=3, seed=0x0A1DE5 + 4 + rank), dev)
        r.reserve(M, n_meshlets)
        mats, _, zn = virtual_shadow_matrices([0.0, 0.0, -60.0], [0.3, -1.0, 0.2], 500.0, 2.0, views)
        lanes = []  # one set of outputs (and one mesh_instances copy: cull_meshes writes lod_index per view) per batch element
        for e in range(min(vb, views)):
            sc = base if e == 0 else dataclasses.replace(base, mesh_instances=base.mesh_instances.clone())
            lanes.append(PreparedFrame.create(sc, with_triangles=False, expand=False))
    groups = []
    for v0 in range(0, views, vb):
        n = min(vb, views - v0)
        cf = (L.PreparedFrame * n)(*[lanes[e].c() for e in range(n)])
        cc = (L.CullGeometryContext * n)()
        for e in range(n):
            cam = base.cull_camera()
            for k in range(16):
                cam.projection_view[k] = float(mats[v0 + e][k])
            cam.position[0], cam.position[1], cam.position[2] = 0.0, 0.0, -60.0
            cam.near_clip = zn
            ctx = CullGeometryContext(init_cull_meshes=True, cull_flags=L.CULL_TEST_FRUSTUM | L.CULL_SELECT_LOD, cull_camera=cam,
                                      stages=L.STAGE_MESHES | L.STAGE_MESHLETS)
            C.memmove(C.byref(cc[e]), C.byref(ctx.c()), C.sizeof(L.CullGeometryContext))
        groups.append((n, cf, cc))

    def check(st):
        if st != L.OXC_OK:
            raise RuntimeError(lib.oxc_last_error(ctxp).decode())

    def one():
        for n, cf, cc in groups:
            check(lib.oxc_cull_geometry(ctxp, cf, cc, sp) if n == 1 else lib.oxc_cull_geometry_batch(ctxp, n, cf, cc, sp))

    with torch.cuda.stream(stream):
        for _ in range(warmup):
            one()
    torch.cuda.synchronize()
    per_view = []
    for n, cf, cc in groups:  # the last step's counters are still in the slots of each element
        for e in range(n):
            out = L.Counters()
            check(lib.oxc_read_counters(ctxp, C.byref(cc[e]), C.byref(out), sp))
            per_view.append((out.total_visible_meshlet_instances, out.cull_triangles_cmd_x))
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
    processed = sum(t for t, _ in per_view)
    if rank == 0:
        print(json.dumps({
            "metric": "meshlets/s culled (meshlets the meshlet stage processed, summed over views)", "value": round(processed * world * steps / dt, 1),
            "unit": "meshlets/s", "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 6), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[4]: {n_meshlets} LOD-0 meshlets x {views} orthographic cascade views, per-view cull_meshes (frustum + LOD select) + cull_meshlets",
                       "meshlets_per_gpu": n_meshlets, "mesh_instances": M, "views": views, "views_per_call": vb,
                       "candidate_meshlet_views_per_step": n_meshlets * views, "processed_meshlet_views_per_step": processed,
                       "per_view_processed": [t for t, _ in per_view], "per_view_visible": [v for _, v in per_view]},
            "roofline": None, "cpu_baseline": None}))
    if dist is not None:
        dist.destroy_process_group()
