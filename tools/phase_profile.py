#!/usr/bin/env python3
"""Per-wave phase breakdown of the intersect kernels on the C2 frame (needs a `make PROF=1` build of the library, e.g.
make -C raytracer_amd/csrc PROF=1 OUT=$PWD/raytracer_amd/librt_engine_prof.so, and RT_ENGINE_LIB pointing at it):
    RT_ENGINE_LIB=raytracer_amd/librt_engine_prof.so python tools/phase_profile.py [build key]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt

key = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0x12345678
scene = rt.default_scene(key)
e = rt.Engine(rt.default_config(width=1920, height=1080, samples_per_axis=8, bounces=8))
e.set_scene(scene)
e.set_camera(rt.camera_look((0.0, 3.0, -10.0), (0.0, 0.0, 1.0)))
e.render_frame(7)
e.stats()          # (prints and clears the counters of the warm-up frame)
print(f"---- build key {key:#x}, one frame", file=sys.stderr)
e.render_frame(7)
st = e.stats()
print("rays", st["rays"], [int(x) for x in st["rays_at_depth"][:9]], "box tests per ray", st["box_tests"] / st["rays"])
