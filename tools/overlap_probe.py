#!/usr/bin/env python3
"""tools/overlap_probe.py -- do the ALU-bound HiZ meshlet test and the HBM-bound triangle stage overlap when they run on two streams?

Two contexts with their own output buffers over the same 10 M-meshlet scene.  A = late meshlet stage (test + emit) of context 1,
B = triangle stage (test + emit) of context 2 over the visible list its own earlier full call left.  Times A alone, B alone,
A then B on one stream, and A || B on two streams.  Experiment tooling."""
import ctypes as C
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
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    K, HW, M = 1000, 4096, 10_000
    N = M * K
    scene = make_scene(SceneSpec(n_mesh_instances=M, meshlets_per_mesh=K, with_geometry=True, seed=0x0A1DE5 + 2), dev)
    depth = ImageAttachment.depth(make_depth(2 * HW, 2 * HW, 64, seed=3, device=dev))
    hiz = ImageAttachment.hiz(HW, HW, dev)
    g = torch.Generator(device=dev).manual_seed(5)
    rs, frames, ctxs = [], [], []
    for k in range(2):
        r = RendererInstance(0)
        r.reserve(M, N)
        f = PreparedFrame.create(scene, with_triangles=True)
        words = f.meshlet_instance_visibility_mask_buffer.numel()
        bits = (torch.rand((words, 32), generator=g, device=dev) < 0.3).to(torch.int64)
        f.meshlet_instance_visibility_mask_buffer.copy_((bits << torch.arange(32, device=dev)).sum(1).to(torch.int32))
        r.prepared_frame = f
        ctx = CullGeometryContext(use_hiz=True, init_cull_meshes=False, cull_flags=L.CULL_TEST_ALL, cull_camera=scene.cull_camera(), hiz_attachment=hiz, stages=L.STAGE_ALL)
        r.seed_meshlet_instances(ctx, N)
        rs.append(r), frames.append(f), ctxs.append(ctx)
    torch.cuda.synchronize()
    mg = L.MainGeometryContext()
    mg.struct_size = C.sizeof(L.MainGeometryContext)
    mg.depth_attachment, mg.hiz_attachment = depth.c(), hiz.c()
    rs[0]._check(rs[0]._lib.oxc_generate_hiz(rs[0]._ctx, C.byref(mg), C.c_void_p(0)))
    torch.cuda.synchronize()
    cf = [f.c() for f in frames]
    cc = [c.c() for c in ctxs]
    for k in range(2):  # a full early + late frame on each context: leaves valid visible lists / counters behind
        for flags in (L.CULL_TEST_ALL, L.CULL_TEST_ALL | L.CULL_LATE_PASS):
            cc[k].cull_flags = flags
            rs[k]._check(rs[k]._lib.oxc_cull_geometry(rs[k]._ctx, C.byref(cf[k]), C.byref(cc[k]), C.c_void_p(0)))
    torch.cuda.synchronize()

    def A(stream):  # late meshlet stage, context 0
        cc[0].cull_flags, cc[0].stages = L.CULL_TEST_ALL | L.CULL_LATE_PASS, L.STAGE_MESHLETS
        rs[0]._check(rs[0]._lib.oxc_cull_geometry(rs[0]._ctx, C.byref(cf[0]), C.byref(cc[0]), C.c_void_p(stream.cuda_stream)))

    big = torch.empty(768 << 20, dtype=torch.uint8, device=dev)  # 0.8 GB streamed: about the late triangle test's traffic

    def B(stream):  # stand-in for the triangle stage: a streaming read of the same volume (the stage cannot run on its own)
        rs[1].stream_read_probe(big, stream=stream)

    def timed(fn, reps=40):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6

    ta = timed(lambda: A(s1))
    tb = timed(lambda: B(s1))
    tab = timed(lambda: (A(s1), B(s1)))

    def both():
        A(s1)
        B(s2)
        # next iteration's A must not overtake this iteration's B and vice versa: join
        e1, e2 = torch.cuda.Event(), torch.cuda.Event()
        e1.record(s1), e2.record(s2)
        s1.wait_event(e2), s2.wait_event(e1)

    tpar = timed(both)
    print(f"A (late meshlet stage) {ta:.1f} us   B (0.8 GB streaming read) {tb:.1f} us   A;B one stream {tab:.1f} us   A||B two streams {tpar:.1f} us")


if __name__ == "__main__":
    main()
