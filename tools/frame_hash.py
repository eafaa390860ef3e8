#!/usr/bin/env python3
"""Print SHA-256 of the C1 and C2 frames of the default scene and their ray counts (a quick way to see that a kernel
change left every byte of the result alone):   python tools/frame_hash.py"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt

scene = rt.default_scene(0x5EED0001)
cam = rt.camera_look((0.0, 3.0, -10.0), (0.0, 0.0, 1.0))
for (W, H, S, B) in ((1280, 720, 3, 3), (1920, 1080, 8, 8)):
    e = rt.Engine(rt.default_config(width=W, height=H, samples_per_axis=S, bounces=B))
    e.set_scene(scene)
    e.set_camera(cam)
    frame = e.render_frame(7)
    st = e.stats()
    print(W, H, S, B, hashlib.sha256(frame.tobytes()).hexdigest()[:32], "rays", st["rays"], [int(x) for x in st["rays_at_depth"][:B + 1]])
