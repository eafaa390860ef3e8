#!/usr/bin/env python3
"""A quarter-million-pixel cut of the 100k-primitive workload (scene read from memory, not LDS):  python tools/c3_quick.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import raytracer_amd as rt

scene, cam, _ = bench.build_scene(rt, "c3")
e = rt.Engine(rt.default_config(width=960, height=540, samples_per_axis=8, bounces=8, flags=int(os.environ.get("RT_FLAGS", "0"))))
e.set_scene(scene)
e.set_camera(cam)
buf = torch.zeros(960 * 540 * 4, dtype=torch.uint8, device="cuda:0")
e.render(7, 0, 1, buf.data_ptr())
e.sync()
e.time_kernels(True)
t0 = time.perf_counter()
for _ in range(3):
    e.render(7, 0, 1, buf.data_ptr())
e.sync()
ms = (time.perf_counter() - t0) / 3 * 1e3
k, n = e.kernel_times()
st = e.stats()
import hashlib
print(f"frame {ms:.1f} ms, intersect {k:.1f} ms in {n} launches, rays {st['rays']}, {st['rays'] / ms / 1e3:.0f} Mrays/s, box tests per ray {st['box_tests'] / st['rays']:.1f}, frame md5 {hashlib.md5(buf.cpu().numpy().tobytes()).hexdigest()}")
