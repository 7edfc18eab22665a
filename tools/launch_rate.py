#!/usr/bin/env python3
"""tools/launch_rate.py -- how many kernel nodes per second a HIP graph replay sustains on this box,
with 1..4 streams inside the graph (trivial kernels: the stream-read probe over 4 KiB)."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oxylus_amd.renderer import RendererInstance  # noqa: E402

dev = torch.device("cuda", 0)
r = RendererInstance(0)
buf = torch.zeros(4096, dtype=torch.uint8, device=dev)
for n_streams in (1, 2, 3, 4):
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
    nodes = 144
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=streams[0]):
        for s in streams[1:]:
            s.wait_stream(streams[0])
        for i in range(nodes):
            r.stream_read_probe(buf, streams[i % n_streams])
        for s in streams[1:]:
            streams[0].wait_stream(s)
    with torch.cuda.stream(streams[0]):
        for _ in range(20):
            g.replay()
    torch.cuda.synchronize()
    reps = 200
    t0 = time.perf_counter()
    with torch.cuda.stream(streams[0]):
        for _ in range(reps):
            g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"streams={n_streams}: {dt / (reps * nodes) * 1e6:.2f} us per trivial kernel node ({nodes} nodes/graph)")
# eager, single stream, from Python
s0 = torch.cuda.Stream(device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20000):
    r.stream_read_probe(buf, s0)
torch.cuda.synchronize()
print(f"eager via ctypes: {(time.perf_counter() - t0) / 20000 * 1e6:.2f} us per launch")
