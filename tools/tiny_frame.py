#!/usr/bin/env python3
"""Render a few tiny frames (fixed costs of the launches dominate) and a C2 frame, printing rays per bounce level.
Run under rocprofv3 --kernel-trace to see what one launch costs when it has next to nothing to do."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt

scene = rt.default_scene(0x5EED0001)
cam = rt.camera_look((0.0, 3.0, -10.0), (0.0, 0.0, 1.0))
for (W, H, S, B) in ((64, 64, 1, 8), (1920, 1080, 8, 8)):
    e = rt.Engine(rt.default_config(width=W, height=H, samples_per_axis=S, bounces=B))
    e.set_scene(scene)
    e.set_camera(cam)
    for _ in range(3):
        e.render(7)
    e.sync()
    st = e.stats()
    print(W, H, S, B, "rays", st["rays"], "per depth", [int(x) for x in st["rays_at_depth"][:B + 1]])
