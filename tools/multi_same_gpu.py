#!/usr/bin/env python3
"""C2 through rt_multi with one GPU listed 1, 2, 3, 4 times: several engines on separate streams share the device, so
the ramp-down of one engine's pass overlaps the body of another's.  Prints ms per frame."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt

W, H, S, B = 1920, 1080, 8, 8
scene = rt.default_scene(0x5EED0001)
cam = rt.camera_look((0.0, 3.0, -10.0), (0.0, 0.0, 1.0))
for n in (1, 2, 3, 4, 1):
    m = rt.MultiEngine(rt.default_config(width=W, height=H, samples_per_axis=S, bounces=B), [0] * n)
    m.set_scene(scene)
    m.set_camera(cam)
    for _ in range(3):
        m.render(7)
    m.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        m.render(7)
    m.sync()
    print(f"{n} engine(s) on one GPU: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per frame", flush=True)
    m.close()
