#!/usr/bin/env python3
"""Debug helper: one random scene of the parity suite through several engine variants against the oracle:
    python tools/dbg_scene.py <seed> [fog 0|1]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # (kept under tests/: it uses the checker)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import raytracer_amd as rt
from conftest import random_tables, build_table_scene
from oracle import orcbind as orc

seed = int(sys.argv[1]); fog = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False
E = rt.engine
verts = E.load_obj(os.path.join(ROOT, "assets", "src", "stb_image", "tree.obj"))
teximg, _ = E.load_png(os.path.join(ROOT, "assets", "src", "stb_image", "tree_texture.png"), 4)
sky, _ = E.load_png(os.path.join(ROOT, "assets", "src", "stb_image", "skybox12.png"), 3)
tex, mat, obj, plane, (pos, look) = random_tables(seed, fog)
args = (tex, mat, obj, plane, verts, teximg, sky, 0xC0FFEE + seed)
scene = build_table_scene(rt.Scene(), *args)
osc = build_table_scene(orc.Scene(orc.MATH_PORTABLE), *args)
print("objects", len(obj), "meshes", int((obj[:, 0] == 3).sum()), scene.flat_info(), scene.regroup_info())
W, H, S, B = 96, 54, 2, 5
cam = rt.camera_look(pos, look)
ref = osc.render(cam, W, H, S, B, seed, threads=8, want=("radiance", "rgba", "rng_draws", "primary_id", "primary_t"))
for name, flags, env in (("default", 0, {}), ("no cull", E.FLAG_NO_CULL, {}), ("no regroup", E.FLAG_NO_REGROUP, {}), ("no lds", E.FLAG_NO_LDS, {}), ("no lds no pairs", E.FLAG_NO_LDS | E.FLAG_NO_PAIRS, {}),
                         ("no lds no order", E.FLAG_NO_LDS | E.FLAG_NO_ORDER, {}), ("no lds no regroup", E.FLAG_NO_LDS | E.FLAG_NO_REGROUP, {})):
    e = rt.Engine(rt.default_config(width=W, height=H, samples_per_axis=S, bounces=B, flags=flags))
    e.set_scene(scene); e.set_camera(cam)
    e.keep_samples(True)
    e.render_frame(seed)
    got = e.read_samples()
    bad = np.nonzero(got["primary_id"].reshape(-1) != ref["primary_id"].reshape(-1))[0]
    if len(bad):
        w_ = ref["primary_id"].reshape(-1); g_ = got["primary_id"].reshape(-1)
        for t_ in np.unique(w_[bad]):
            sel = np.nonzero(w_ == t_)[0]
            print(f"   wanted {t_}: {len(sel)} samples in all, {np.isin(sel, bad).sum()} wrong; got instead {np.unique(g_[sel], return_counts=True)}; pixels of the wrong ones {np.unique(bad[np.isin(bad, sel)] // (S * S))[:12]}")
    print(f"{name:20s} ids differ: {len(bad)}", "" if not len(bad) else f"first {bad[:4]} got {got['primary_id'].reshape(-1)[bad[:4]]} want {ref['primary_id'].reshape(-1)[bad[:4]]}; wanted ids {np.unique(ref['primary_id'].reshape(-1)[bad])}")
    e.close()
for i in np.unique(ref["primary_id"]):
    if i >= 0: print("object", i, obj[i])

# rays at one object through the variants
if len(sys.argv) > 3:
    t = int(sys.argv[3])
    rng = np.random.default_rng(1)
    n = 64
    if obj[t, 0] == 1: p = rng.uniform(obj[t, 1:4], obj[t, 4:7], (n, 3))
    else: p = obj[t, 1:4] + rng.normal(size=(n, 3)) * 0.3 * abs(obj[t, 4])
    o = np.tile(np.array(pos, np.float32), (n, 1))
    d = (p - o).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    want = osc.trace(o, d)[0]
    print("oracle ids", np.unique(want, return_counts=True))
    for name, flags in (("default", 0), ("no cull", E.FLAG_NO_CULL), ("no pairs", E.FLAG_NO_PAIRS), ("no regroup", E.FLAG_NO_REGROUP), ("no cull no regroup", E.FLAG_NO_CULL | E.FLAG_NO_REGROUP)):
        e = rt.Engine(rt.default_config(width=W, height=H, samples_per_axis=S, bounces=B, flags=flags))
        e.set_scene(scene); e.set_camera(cam)
        got = e.trace(o, d)[0]
        print(f"{name:20s} differ {(got != want).sum()} of {n}; got ids {np.unique(got, return_counts=True)}")
        e.close()
