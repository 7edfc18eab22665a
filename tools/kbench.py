#!/usr/bin/env python3
"""tools/kbench.py -- A/B several builds of liboxcull.so on the configs[2] frame in ONE process (one scene generation).

  python tools/kbench.py [--libs base=oxylus_amd/liboxcull.so,x=oxylus_amd/variants/liboxcull_x.so] [--frames 60] [--meshlets N]
  A library entry may carry settings: tag=path@ASYNC=1@SHARE=1@UNORD=1@TUNE0=3 -- ASYNC=1 sets async_triangles, SHARE=1 share_pass_tests, UNORD=n
  unordered_output on every call (lists are then compared as sorted sets), TUNEk=v calls oxc_debug_set_tuning(k, v).

For every library: warm up, time `--frames` frames (wall, one stream), then an instrumented pass (HIP-event pair per kernel), and a
checksum of every output of one frame (visible lists, packed indices, mask, pyramid) -- variants must agree with the first library
bit for bit, or the line says MISMATCH.  Experiment tooling; bench.py is the measurement of record."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oxylus_amd import lib as L  # noqa: E402
from oxylus_amd.renderer import CullGeometryContext, ImageAttachment, PreparedFrame, RendererInstance  # noqa: E402
from oxylus_amd.synth import SceneSpec, make_depth, make_scene  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="base=oxylus_amd/liboxcull.so")
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--meshlets", type=int, default=10_000_000)
    ap.add_argument("--out", default="")
    ap.add_argument("--tris", type=int, default=64, help="triangles per meshlet; > 64: wide_triangle_index (at most 8M meshlets)")
    ap.add_argument("--coherent-mask", action="store_true", help="prior-visibility mask = the previous frame's result instead of random p = 0.3")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream(device=dev)
    K, HW = 1000, 4096
    wide = a.tris > 64
    if wide:
        a.meshlets = min(a.meshlets, 8_000_000)
    M = a.meshlets // K
    N = M * K
    with torch.cuda.stream(stream):
        scene = make_scene(SceneSpec(n_mesh_instances=M, meshlets_per_mesh=K, with_geometry=True, seed=0x0A1DE5 + 2, tris_per_meshlet=a.tris), dev)
        frame = PreparedFrame.create(scene, with_triangles=True, max_tris=128 if wide else 64)
        depth = ImageAttachment.depth(make_depth(2 * HW, 2 * HW, 64, seed=3, device=dev))
        hiz = ImageAttachment.hiz(HW, HW, dev)
        g = torch.Generator(device=dev).manual_seed(5)
        words = frame.meshlet_instance_visibility_mask_buffer.numel()
        bits = (torch.rand((words, 32), generator=g, device=dev) < 0.3).to(torch.int64)
        mask0 = (bits << torch.arange(32, device=dev)).sum(1).to(torch.int32)
        del bits
    torch.cuda.synchronize()
    mask = frame.meshlet_instance_visibility_mask_buffer
    results, ref_sum = [], None
    for item in a.libs.split(","):
        tag, path = item.split("=", 1)
        path, *settings = path.split("@")
        use_async, use_share, unord, tunes = False, False, 0, []
        for kv in settings:
            k_, v_ = kv.split("=")
            if k_ == "ASYNC":
                use_async = v_ == "1"
            elif k_ == "SHARE":
                use_share = v_ == "1"
            elif k_ == "UNORD":
                unord = int(v_)
            elif k_.startswith("TUNE"):
                tunes.append((int(k_[4:]), int(v_)))
            else:
                raise SystemExit(f"unknown setting {kv}")
        r = RendererInstance(0, lib_path=os.path.join(ROOT, path) if not os.path.isabs(path) else path)
        for knob, val in tunes:
            r.debug_set_tuning(knob, val)
        lib, ctxp, sp = r._lib, r._ctx, C.c_void_p(stream.cuda_stream)
        r.reserve(M, N)
        r.prepared_frame = frame
        ctx = CullGeometryContext(use_hiz=True, init_cull_meshes=False, cull_flags=L.CULL_TEST_ALL, cull_camera=scene.cull_camera(), hiz_attachment=hiz, stages=L.STAGE_ALL,
                                  wide_triangle_index=wide)
        with torch.cuda.stream(stream):
            r.seed_meshlet_instances(ctx, N)
        cframe, cctx = frame.c(), ctx.c()
        cctx.async_triangles = int(use_async)
        cctx.unordered_output = unord
        if use_share:
            cctx.share_pass_tests = 1
        mg = L.MainGeometryContext()
        mg.struct_size = C.sizeof(L.MainGeometryContext)
        mg.depth_attachment, mg.hiz_attachment = depth.c(), hiz.c()

        def check(st):
            if st != L.OXC_OK:
                raise RuntimeError(lib.oxc_last_error(ctxp).decode())

        sums = []

        def one(record=False):
            mask.copy_(mask0, non_blocking=True)
            check(lib.oxc_generate_hiz(ctxp, C.byref(mg), sp))
            for flags in (L.CULL_TEST_ALL, L.CULL_TEST_ALL | L.CULL_LATE_PASS):
                cctx.cull_flags = flags
                check(lib.oxc_cull_geometry(ctxp, C.byref(cframe), C.byref(cctx), sp))
                if record:
                    torch.cuda.synchronize()
                    out = L.Counters()
                    check(lib.oxc_read_counters(ctxp, C.byref(cctx), C.byref(out), sp))
                    first = out.early_visible_meshlet_instances if flags & L.CULL_LATE_PASS else 0
                    vis = frame.visible_meshlet_instances_indices_buffer[first:first + out.cull_triangles_cmd_x].to(torch.int64)
                    idx = frame.reordered_indices_buffer[:out.draw_index_count].to(torch.int64) & 0xFFFFFFFF
                    if unord:  # the ordered form's lists ascend: sorted, an unordered list must be those bytes
                        vis, idx = torch.sort(vis)[0], torch.sort(idx)[0]
                    w = torch.arange(1, 1 + vis.numel(), device=dev, dtype=torch.int64)
                    sums.append((out.cull_triangles_cmd_x, out.draw_index_count, int((vis * (w % 1000003)).sum().item()),
                                 int((idx * (torch.arange(1, 1 + idx.numel(), device=dev, dtype=torch.int64) % 1000003)).sum().item())))
            if record:
                sums.append(int(mask.to(torch.int64).sum().item()))
                sums.append(int(hiz.data.view(torch.int32).to(torch.int64).sum().item()))

        with torch.cuda.stream(stream):
            one(record=True)
            for _ in range(10):
                one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            for _ in range(a.frames):
                one()
        torch.cuda.synchronize()
        wall_us = (time.perf_counter() - t0) / a.frames * 1e6
        r.profile_begin()
        with torch.cuda.stream(stream):
            for _ in range(a.frames):
                one()
        p = r.profile_end()
        ks = {k: round(v["total_ms"] / v["launches"] * 1e3, 2) for k, v in p["kernels"].items()}
        per_frame = {k: round(v["total_ms"] / a.frames * 1e3, 2) for k, v in p["kernels"].items()}
        if ref_sum is None:
            ref_sum = sums
        ok = sums == ref_sum
        results.append({"tag": tag, "frame_us": round(wall_us, 1), "kernel_sum_us": round(sum(per_frame.values()), 1), "match": ok, "kernels_avg_us": ks})
        print(f"{tag:>16s}  frame {wall_us:8.1f} us  sum {sum(per_frame.values()):8.1f}  {'ok' if ok else 'MISMATCH'}  " +
              "  ".join(f"{k.replace('cull_', '')}={v}" for k, v in ks.items()), flush=True)
        r.close()
    if a.out:
        with open(a.out, "w") as f:
            json.dump({"meshlets": N, "frames": a.frames, "results": results}, f, indent=1)


if __name__ == "__main__":
    main()
