#!/usr/bin/env python3
"""C2 frame time and the part of it spent in the intersect kernels (HIP events around those launches; the rest is shading
and the resolve):   python tools/pass_times.py     (RT_ENGINE_LIB=<other build> for A/B runs on one box)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt

scene = rt.default_scene(int(os.environ.get("RT_BUILD_KEY", "0x5EED0001"), 0))   # RT_BUILD_KEY: another BVH build key
e = rt.Engine(rt.default_config(width=1920, height=1080, samples_per_axis=8, bounces=8))
e.set_scene(scene)
e.set_camera(rt.camera_look((0.0, 3.0, -10.0), (0.0, 0.0, 1.0)))
buf = torch.zeros(1920 * 1080 * 4, dtype=torch.uint8, device="cuda:0")
for _ in range(3):
    e.render(7, 0, 1, buf.data_ptr())
e.sync()
e.time_kernels(True)
t0 = time.perf_counter()
for _ in range(10):
    e.render(7, 0, 1, buf.data_ptr())
e.sync()
ms = (time.perf_counter() - t0) / 10 * 1e3
k, n = e.kernel_times()
print(f"frame {ms:.2f} ms, intersect {k:.2f} ms in {n} launches, rest {ms - k:.2f} ms")
