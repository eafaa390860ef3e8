#!/usr/bin/env python3
"""Throughput with K frames in flight on one GPU: K engines (own stream and queues each) take the frames of a sequence in
turn, nobody waits between frames.   python tools/frames_in_flight.py [n_ranks [k ...]]   (rank 0's share of the C2 frame)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt

n_ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 1
W, H, S, B = 1920, 1080, 8, 8
scene = rt.default_scene(0x5EED0001)
cam = rt.camera_look((0.0, 3.0, -10.0), (0.0, 0.0, 1.0))
for k in [int(a) for a in sys.argv[2:]] or [1, 2, 3, 1]:
    engs, bufs = [], []
    for _ in range(k):
        e = rt.Engine(rt.default_config(width=W, height=H, samples_per_axis=S, bounces=B))
        e.set_scene(scene)
        e.set_camera(cam)
        engs.append(e)
        bufs.append(torch.zeros(max(e.tile_bytes(n_ranks), W * H * 4), dtype=torch.uint8, device="cuda:0"))
    for f in range(2 * k):
        engs[f % k].render(7, 0, n_ranks, bufs[f % k].data_ptr())
    for e in engs:
        e.sync()
    n = 24
    t0 = time.perf_counter()
    for f in range(n):
        engs[f % k].render(7, 0, n_ranks, bufs[f % k].data_ptr())
    for e in engs:
        e.sync()
    print(f"n_ranks={n_ranks}, {k} frame(s) in flight: {(time.perf_counter() - t0) / n * 1e3:.2f} ms per frame", flush=True)
    for e in engs:
        e.close()
