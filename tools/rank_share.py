#!/usr/bin/env python3
"""Time one rank's share of the C2 frame (rank 0 of N, tiles only, no gather) on one GPU:  python tools/rank_share.py [N ...]
What a rank of an N-GPU job spends per frame, to see how the fixed costs of the passes limit strong scaling."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt

scene = rt.default_scene(0x5EED0001)
e = rt.Engine(rt.default_config(width=1920, height=1080, samples_per_axis=8, bounces=8))
e.set_scene(scene)
e.set_camera(rt.camera_look((0.0, 3.0, -10.0), (0.0, 0.0, 1.0)))
for n in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    buf = torch.zeros(max(e.tile_bytes(n), 1920 * 1080 * 4), dtype=torch.uint8, device="cuda:0")
    for _ in range(2):
        e.render(7, 0, n, buf.data_ptr())
    e.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        e.render(7, 0, n, buf.data_ptr())
    e.sync()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print(f"n_ranks={n}: {ms:.2f} ms per frame share, ideal {1.0 / n:.3f} of n=1")
