"""TEST INFRASTRUCTURE: writes tests/golden/scene_c3.npz — the 100 000-primitive traversal-stress scene at its real
size — from the REFERENCE ITSELF (oracle/_ref/libref.so: the reference's unmodified C++ behind oracle/ref_harness.cpp;
`make -C oracle ref` first).  Runs only where /root/reference exists.  Every vector is also compared with the C
restatement (oracle/rt_oracle.c, libm math mode) and any mismatch aborts.

The scene: raytracer_amd.synthetic.synthetic_tables(100000) (spheres and boxes from one PCG32 stream), the default metal
checker plane and skybox, camera (0,20,-150) looking along +z, built by BVH_Node::BVH_Node (Object.cpp:181-225) under
build key BUILD_KEY + 2.  Stored:
  * sha256 of the inner-node boxes, left links, right links and first-visit leaf order of the reference's tree, plus
    every 1000th row of each for a readable diagnosis;
  * sample (0,0) of every pixel of a 480 x 270 frame: primary-hit ids and HitInfo.distance bits;
  * a 32 x 32 crop of the 3840 x 2160, 256 spp, 8 bounce frame: ids, distances, RNG draw counts and 8-bit colour of
    every sample, float radiance of its top-left 16 x 16 pixels.

    python oracle/gen_golden_c3.py        # ~2 min on 8 cores, most of it the reference's O(n^2) tree build
"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import orcbind as O  # noqa: E402
from oracle.gen_golden import BUILD_KEY, SEED, must, save  # noqa: E402
from oracle.refbind import Ref  # noqa: E402
from raytracer_amd.synthetic import build_from_tables, synthetic_tables  # noqa: E402

N = 100000
CAM = ((0.0, 20.0, -150.0), (0.0, 0.0, 1.0))
FULL = (3840, 2160, 16, 8)
CROP = (1904, 1064, 1936, 1096)      # 32 x 32 pixels around the image centre: the densest part of the object cloud
MAP = (480, 270, 1, 8)


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def main():
    t_all = time.time()
    R = Ref()
    sky, _ = R.load_image("src/stb_image/skybox12.png", 3)
    obj, tex, mat = synthetic_tables(N)
    plane = [[0.0, len(mat) - 1]]
    key = BUILD_KEY + 2
    w = R.world(BUILD_KEY)
    t0 = time.time()
    R.set_custom(w, key, tex, mat, obj, plane)
    t_ref_build = time.time() - t0
    print(f"  reference built the {N}-primitive tree in {t_ref_build:.1f} s")
    t0 = time.time()
    s = build_from_tables(O.Scene(O.MATH_LIBM), tex, mat, obj, plane, sky, key)
    print(f"  oracle built it in {time.time() - t0:.1f} s")
    top = R.dump_top(w)
    for name, a, b in zip(("boxes", "left", "right", "leaf"), top, s.dump_top()):
        must("c3 top " + name, a, b)
    cam = O.camera(*CAM)
    R.set_camera(w, *CAM)
    Wm, Hm, Sm, Bm = MAP
    a = R.render(w, Wm, Hm, Sm, Bm, SEED + 2, threads=8, want=("primary_id", "primary_t"))
    b = s.render(cam, Wm, Hm, Sm, Bm, SEED + 2, threads=8, want=("primary_id", "primary_t"))
    must("c3 map id", a["primary_id"], b["primary_id"])
    must("c3 map t", a["primary_t"], b["primary_t"])
    W, H, S, B = FULL
    want = ("radiance", "primary_id", "primary_t", "rng_draws", "rgba")
    t0 = time.time()
    ac = R.render(w, W, H, S, B, SEED + 2, rect=CROP, threads=8, want=want)
    print(f"  reference rendered the crop in {time.time() - t0:.1f} s")
    bc = s.render(cam, W, H, S, B, SEED + 2, rect=CROP, threads=8, want=want)
    for k in ac:
        must("c3 crop " + k, ac[k], bc[k])
    cw = CROP[2] - CROP[0]
    sub = (np.arange(16)[:, None] * cw + np.arange(16)[None, :]).reshape(-1)      # top-left 16 x 16 pixels of the crop
    save("scene_c3", n=np.array(N), build_key=np.array(key, np.uint32), seed=np.array(SEED + 2), cam_pos=np.array(CAM[0], np.float32),
         cam_look=np.array(CAM[1], np.float32), n_inner=np.array(len(top[1])),
         top_boxes_sha256=sha(top[0]), top_left_sha256=sha(top[1]), top_right_sha256=sha(top[2]), top_leaf_ids_sha256=sha(top[3]),
         top_boxes_rows=top[0][::1000].copy(), top_left_rows=top[1][::1000].copy(), top_right_rows=top[2][::1000].copy(),
         top_leaf_ids_rows=top[3][::1000].copy(),
         map_cfg=np.array(MAP), map_primary_id=a["primary_id"].reshape(Hm, Wm), map_primary_t=a["primary_t"].reshape(Hm, Wm),
         cfg=np.array(FULL), crop=np.array(CROP), crop_primary_id=ac["primary_id"], crop_primary_t=ac["primary_t"],
         crop_rng_draws=ac["rng_draws"].astype(np.uint16), crop_rgba=ac["rgba"], crop_radiance_pixels=sub,
         crop_radiance=ac["radiance"][sub].copy(), reference_build_seconds=np.array(t_ref_build))
    assert int(ac["rng_draws"].max()) < 65536
    R.free(w)
    print(f"scene_c3 written; the oracle restatement matched the reference throughout ({time.time() - t_all:.0f} s)")


if __name__ == "__main__":
    main()
