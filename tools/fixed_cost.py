#!/usr/bin/env python3
"""Frame and intersect time of the default scene at 1920 x H for several H:  python tools/fixed_cost.py [bounces] [H ...]
Shows the cost a pass has however few rays it holds (what limits strong scaling over ranks)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt

scene = rt.default_scene(0x5EED0001)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for H in [int(a) for a in sys.argv[2:]] or [1080, 540, 270, 136, 64]:
    e = rt.Engine(rt.default_config(width=1920, height=H, samples_per_axis=8, bounces=B))
    e.set_scene(scene)
    e.set_camera(rt.camera_look((0.0, 3.0, -10.0), (0.0, 0.0, 1.0)))
    buf = torch.zeros(1920 * H * 4, dtype=torch.uint8, device="cuda:0")
    for _ in range(2):
        e.render(7, 0, 1, buf.data_ptr())
    e.sync()
    e.stats()
    e.time_kernels(True)
    t0 = time.perf_counter()
    for _ in range(5):
        e.render(7, 0, 1, buf.data_ptr())
    e.sync()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    k, n = e.kernel_times()
    st = e.stats()
    print(f"B={B} H={H}: frame {ms:.3f} ms, intersect {k:.3f} ms in {n} launches, rays {st['rays']}, {st['rays'] / ms / 1e3:.0f} Mrays/s", flush=True)
    del e
