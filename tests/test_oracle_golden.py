"""The CPU oracle (oracle/rt_oracle.c, libm math like the reference) against the golden vectors that the reference's own
C++ produced (oracle/gen_golden.py).  Everything here is bit-for-bit.  No GPU."""
import hashlib

import numpy as np
import pytest

from conftest import assert_bits, build_table_scene, synthetic_tables

CAM_POS, CAM_LOOK = (0.0, 3.0, -10.0), (0.0, 0.0, 1.0)


def test_rng_known_answers(orc, golden):
    g = golden("rng")
    L = orc.lib()
    got = np.array([[L.orc_rand31(int(k), c) for c in range(16)] for k in g["keys"]], np.uint32)
    assert_bits(got, g["rand31"], "rand31")
    px = [int(p) for p in g["sample_key_pixels"]]
    got = np.array([[L.orc_sample_key(int(g["sample_key_seed"]), p, s) for s in range(4)] for p in px], np.uint32)
    assert_bits(got, g["sample_keys"], "sample keys")
    # Random01 / RandomUnitDistance (DataUtility.h:11-12) from the raw draws
    raw = np.array([L.orc_rand31(0xDEADBEEF, c) for c in range(32)], np.int64)
    assert_bits((raw.astype(np.float32) / np.float32(2147483648.0)).astype(np.float32), g["random01"], "Random01")
    assert_bits(((raw - 1073741823).astype(np.float32) / np.float32(1073741824.0)).astype(np.float32), g["random_unit_distance"], "RandomUnitDistance")
    assert g["random_unit_distance"].min() >= -1.0 and g["random_unit_distance"].max() <= 1.0


def test_aabb_known_answers_from_main_cpp(orc, golden):
    """the 14-vector table commented out in the reference's Main.cpp:24-46"""
    g = golden("aabb")
    dn = orc.normalize(g["kat_d"])
    assert_bits(dn, g["kat_d_normalized"], "glm::normalize")
    got = orc.aabb_hit(g["box"], g["kat_o"], dn)
    assert_bits(got, g["kat_reference"], "DoesRayHit on the Main.cpp table")
    assert_bits(got, g["kat_expect_comment"], "the table's own 'should hit' column")


def test_aabb_random_and_degenerate(orc, golden):
    g = golden("aabb")
    for b, ref in zip(g["boxes"], g["hits"]):
        assert_bits(orc.aabb_hit(b, g["rays_o"], g["rays_d"]), ref, f"DoesRayHit box {b}")


@pytest.mark.parametrize("name,kind", [("sphere", 0), ("sphere_negr", 0), ("cube", 1), ("plane", 4), ("plane_y", 4)])
def test_primitive_intersect(orc, golden, name, kind):
    g = golden("prims")
    hit, rec = orc.intersect_prim(kind, g[name + "_params"], g["rays_o"], g["rays_d"], float(g["fill"]))
    assert_bits(hit, g[name + "_hit"], name + " hit flags")
    assert_bits(rec, g[name + "_rec"], name + " HitInfo fields (untouched fields keep the fill pattern)")


@pytest.mark.parametrize("name,alpha", [("tri", False), ("tri_alpha", True), ("tri_degenerate", False)])
def test_triangle_intersect(orc, golden, assets, name, alpha):
    g = golden("prims")
    hit, rec = orc.intersect_tri(g[name + "_verts"], assets["tex"] if alpha else None, g["tri_rays_o"], g["tri_rays_d"], float(g["fill"]))
    assert_bits(hit, g[name + "_hit"], name + " hit flags")
    assert_bits(rec, g[name + "_rec"], name + " HitInfo fields")
    if name == "tri_degenerate":
        assert (hit == -1).all()     # denominator == 0 -> miss (Object.cpp:530-532)
    if name == "tri_alpha":
        assert (g["tri_hit"] >= 0).sum() > (hit >= 0).sum() > 0   # the cut-out removes some hits


def test_default_scene_structure(golden, oracle_default_libm):
    g = golden("scene_default")
    s = oracle_default_libm
    for a, n in zip(s.dump_top(), ("top_boxes", "top_left", "top_right", "top_leaf_ids")):
        assert_bits(a, g[n], n)
    for a, n in zip(s.dump_mesh(2), ("mesh_boxes", "mesh_left", "mesh_right", "mesh_tris")):
        assert_bits(a, g[n], n)
    assert_bits(s.dump_objects(19), g["objects"], "object parameters and materials")
    assert_bits(s.dump_instance(2), g["instance"], "tree instance transform")
    # the hazards the default scene contains: negative radii -> inverted boxes
    radii = g["objects"][3:18, 4]
    assert (radii < 0).sum() == 5


def test_camera(orc, golden):
    g = golden("scene_default")
    assert_bits(orc.camera(CAM_POS, CAM_LOOK), g["camera"], "Camera::SetForwardVector")
    path = golden("camera_path")
    cam = orc.camera(CAM_POS, CAM_LOOK)
    for f in range(240):
        cam = orc.camera_input(cam, 3.0, 1.0, w=int(f < 120), d=int(f >= 120))
        assert_bits(cam, path["cams"][f], f"camera after scripted frame {f}")


def test_trace_and_ray_color(golden, oracle_default_libm):
    g = golden("trace_default")
    s = oracle_default_libm
    hit, rec = s.trace(g["rays_o"], g["rays_d"])
    assert_bits(hit, g["hit"], "closest-hit ids")
    rec[hit == 18, 7:] = 0          # the plane never writes uv
    assert_bits(rec, g["rec"], "HitInfo after the closest-hit search (stale fields included)")
    n = len(g["keys"])
    assert_bits(s.ray_color(g["rays_o"][:n], g["rays_d"][:n], int(g["ray_color_bounces"]), g["keys"]), g["ray_color"], "GetRayColor")
    assert_bits(s.skybox_lookup(g["sky_dirs"]), g["sky_rgb"], "GetSkyboxPixel")


def test_surface_rays_and_stale_hitinfo(golden, oracle_default_libm):
    """rays leaving sphere surfaces; for the BVH's leftmost leaf Sphere::Intersect assigns a negative distance before it
    rejects (Object.cpp:17-19), the plane then loses the comparison of World.cpp:114 and the ray ends in the sky"""
    g = golden("trace_default")
    s = oracle_default_libm
    hit, rec = s.trace(g["surf_o"], g["surf_d"])
    assert_bits(hit, g["surf_hit"], "surface-ray ids")
    rec[hit == 18, 7:] = 0
    assert_bits(rec, g["surf_rec"], "surface-ray HitInfo")
    stale = (hit == -1) & (rec[:, 0] != np.finfo(np.float32).max)
    assert stale.sum() > 100 and (rec[stale, 0] < 0).all()
    # those rays point at the floor often enough that ignoring the quirk would change ids
    t_plane = -(g["surf_o"][stale, 1] - np.float32(0.001)) / g["surf_d"][stale, 1]
    assert (t_plane >= 0.001).sum() > 10
    col = s.ray_color(g["surf_o"], g["surf_d"], int(g["surf_color_bounces"]), g["surf_keys"])
    assert_bits(col, g["surf_color"], "GetRayColor of surface rays")


def test_c1_frame_rows_and_crop(golden, oracle_default_libm):
    g = golden("frame_c1")
    W, H, S, B = (int(v) for v in g["cfg"])
    seed = int(g["seed"])
    cam = g_cam = None
    from oracle import orcbind as O
    cam = O.camera(CAM_POS, CAM_LOOK)
    s = oracle_default_libm
    crop = tuple(int(v) for v in g["crop"])
    out = s.render(cam, W, H, S, B, seed, rect=crop, want=("radiance", "primary_t", "ray", "rng_draws"))
    assert_bits(out["radiance"], g["crop_radiance"], "crop radiance")
    assert_bits(out["primary_t"], g["crop_primary_t"], "crop distances")
    assert_bits(out["ray"], g["crop_ray"], "crop camera rays")
    assert_bits(out["rng_draws"], g["crop_rng_draws"], "crop RNG draws")
    band = (0, 320, W, 400)
    out = s.render(cam, W, H, S, B, seed, rect=band, threads=8, want=("primary_id", "rgba"))
    assert_bits(out["rgba"], g["rgba"][320:400], "frame rows 320..400")
    assert_bits(out["primary_id"].astype(np.int8), g["primary_id"].reshape(H, W, S * S)[320:400].reshape(-1, S * S), "ids rows 320..400")
    # the reference's own CalculateColorForScreenPosition under one key per pixel
    y0, y1 = (int(v) for v in g["pixel_keyed_rows"])
    assert_bits(s.render_pixel_keyed(cam, W, H, S, B, seed, (0, y0, W, y1)), g["pixel_keyed_rgba"], "pixel-keyed rows")


def test_framemanager_frame(golden, oracle_default_libm):
    """the frame the reference's own scheduler produced (7 workers, stride 37) equals the restated scheduler's"""
    g = golden("frame_c1_framemanager")
    c1 = golden("frame_c1")
    W, H, S, B = (int(v) for v in c1["cfg"])
    from oracle import orcbind as O
    rgba = oracle_default_libm.render_framemanager(O.camera(CAM_POS, CAM_LOOK), W, H, S, B)
    assert hashlib.sha256(rgba.tobytes()).digest() == g["rgba_sha256"].tobytes()
    assert_bits(rgba[::90], g["rows"], "sampled rows")
    # worker pixel order covers its range exactly once (FrameManager.cpp:80-107)
    order = O.thread_pixel_order(W, H, 7, 37, 3)
    lo, hi = 3 * W * H // 7, 4 * W * H // 7
    assert len(order) == hi - lo and np.array_equal(np.sort(order), np.arange(lo, hi, dtype=np.uint32))
    assert order[1] - order[0] == 37


def test_c2_crops_and_thumbnail(golden, oracle_default_libm):
    g = golden("frame_c2")
    W, H, S, B = (int(v) for v in g["cfg"])
    from oracle import orcbind as O
    cam = O.camera(CAM_POS, CAM_LOOK)
    s = oracle_default_libm
    for i in range(3):
        rect = tuple(int(v) for v in g[f"crop{i}"])
        out = s.render(cam, W, H, S, B, int(g["seed"]), rect=rect, threads=8, want=("radiance", "primary_id", "primary_t", "rng_draws", "rgba"))
        assert_bits(out["radiance"], g[f"crop{i}_radiance"], f"crop{i} radiance")
        assert_bits(out["primary_id"].astype(np.int8), g[f"crop{i}_primary_id"], f"crop{i} ids")
        assert_bits(out["primary_t"], g[f"crop{i}_primary_t"], f"crop{i} distances")
        assert_bits(out["rng_draws"].astype(np.uint16), g[f"crop{i}_rng_draws"], f"crop{i} draws")
        assert_bits(out["rgba"], g[f"crop{i}_rgba"], f"crop{i} colour")
    Wt, Ht, St, Bt = (int(v) for v in g["thumb_cfg"])
    out = s.render(cam, Wt, Ht, St, Bt, int(g["seed"]), threads=8, want=("primary_id", "rgba"))
    assert_bits(out["primary_id"].astype(np.int8), g["thumb_primary_id"], "thumbnail ids")
    assert_bits(out["rgba"], g["thumb_rgba"], "thumbnail colour")
    rows = s.render(cam, W, H, 1, B, int(g["seed"]), rect=(0, 512, W, 520), want=("primary_id",))
    assert_bits(rows["primary_id"].astype(np.int8).reshape(8, W), g["s0_primary_id"][512:520], "sample-0 ids rows 512..520")


def test_custom_scene(orc, golden, assets):
    g = golden("scene_custom")
    s = build_table_scene(orc.Scene(orc.MATH_LIBM), g["tex"], g["mat"], g["obj"], g["plane"], assets["verts"], assets["tex"], assets["sky"], int(g["build_key"]))
    W, H, S, B = (int(v) for v in g["cfg"])
    out = s.render(orc.camera(g["cam_pos"], g["cam_look"]), W, H, S, B, int(g["seed"]), threads=8)
    assert_bits(out["primary_id"].astype(np.int8), g["primary_id"], "custom ids")
    assert_bits(out["primary_t"], g["primary_t"], "custom distances")
    assert_bits(out["radiance"], g["radiance"], "custom radiance (isotropic, emissive, metal, image and checker materials)")
    assert_bits(out["rgba"], g["rgba"], "custom frame")


def test_synthetic_scene(orc, golden, assets):
    g = golden("scene_synthetic")
    n = int(g["n"])
    obj, tex, mat = synthetic_tables(n)
    s = build_table_scene(orc.Scene(orc.MATH_LIBM), tex, mat, obj, [[0.0, len(mat) - 1]], None, None, assets["sky"], int(g["build_key"]))
    top = s.dump_top()
    assert hashlib.sha256(top[0].tobytes()).digest() == g["top_boxes_sha256"].tobytes()
    assert_bits(top[1], g["top_left"], "BVH left links")
    assert_bits(top[2], g["top_right"], "BVH right links")
    assert_bits(top[3], g["top_leaf_ids"], "BVH leaf order (std::sort restated)")
    W, H, S, B = (int(v) for v in g["cfg"])
    out = s.render(orc.camera((0.0, 20.0, -150.0), (0.0, 0.0, 1.0)), W, H, S, B, int(g["seed"]), threads=8, want=("primary_id", "primary_t", "rgba"))
    assert_bits(out["primary_id"].astype(np.int16), g["primary_id"], "synthetic ids")
    assert_bits(out["primary_t"].reshape(H, W, S * S)[::6], g["primary_t_rows"], "synthetic distances")
    assert_bits(out["rgba"], g["rgba"], "synthetic frame")


def test_portable_math_is_close_to_libm(orc):
    """the fp64-evaluated sinf / atan2f the kernels use: within half an ulp of the exact value almost everywhere"""
    L = orc.lib()
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-100, 100, 20000), rng.uniform(-2e4, 2e4, 5000)]).astype(np.float32)
    s = np.array([L.orc_pm_sinf(float(v)) for v in x], np.float32)
    exact = np.sin(x.astype(np.float64))
    assert np.mean(s == exact.astype(np.float32)) > 0.9999
    y = rng.normal(size=20000).astype(np.float32)
    z = rng.normal(size=20000).astype(np.float32)
    a = np.array([L.orc_pm_atan2f(float(p), float(q)) for p, q in zip(y, z)], np.float32)
    assert np.mean(a == np.arctan2(y.astype(np.float64), z.astype(np.float64)).astype(np.float32)) > 0.9999
    assert L.orc_pm_atan2f(0.0, -1.0) == np.float32(np.pi) and L.orc_pm_atan2f(-0.0, -1.0) == -np.float32(np.pi)
    assert L.orc_pm_atan2f(0.0, 0.0) == 0.0 and L.orc_pm_atan2f(1.0, 0.0) == np.float32(np.pi / 2)
    assert np.isnan(L.orc_pm_sinf(float("inf")))
    assert L.orc_pm_sinf(-0.0) == 0.0 and L.orc_pm_sinf(0.0) == 0.0


def test_portable_vs_libm_frames_differ_by_little(orc, golden, assets, build_key, oracle_default_libm, oracle_default):
    """same scene, same keys: swapping libm's sinf/atan2f for the portable ones changes almost no byte"""
    cam = orc.camera(CAM_POS, CAM_LOOK)
    a = oracle_default_libm.render(cam, 320, 180, 3, 3, 5, threads=8, want=("primary_id", "rgba", "rng_draws"))
    b = oracle_default.render(cam, 320, 180, 3, 3, 5, threads=8, want=("primary_id", "rgba", "rng_draws"))
    assert_bits(a["primary_id"], b["primary_id"], "ids do not depend on sinf/atan2f")
    assert_bits(a["rng_draws"], b["rng_draws"], "path geometry does not depend on sinf/atan2f")
    assert np.mean(a["rgba"] == b["rgba"]) > 0.9995


def test_std_sort_restatement_sorts(orc):
    rng = np.random.default_rng(1)
    for n in (0, 1, 2, 16, 17, 1000, 50000):
        keys = rng.integers(0, max(1, n // 4 + 1), n).astype(np.float32)      # many ties
        order = orc.std_sort(keys)
        assert np.array_equal(np.sort(order), np.arange(n))
        assert (np.diff(keys[order]) >= 0).all()
    # an organ-pipe input drives introsort towards its depth limit (heapsort fallback path)
    n = 1 << 16
    keys = np.concatenate([np.arange(n // 2), np.arange(n // 2)[::-1]]).astype(np.float32)
    order = orc.std_sort(keys)
    assert (np.diff(keys[order]) >= 0).all()


def test_fog_scene(orc, golden, assets):
    """the reference's alternative scene (World.cpp:34-40) plus more fog volumes: Fog::Intersect draws random numbers during
    the traversal, so ids, distances and radiance only match if every leaf is tested in the reference's order"""
    g = golden("scene_fog")
    s = build_table_scene(orc.Scene(orc.MATH_LIBM), g["tex"], g["mat"], g["obj"], g["plane"], None, None, assets["sky"], int(g["build_key"]))
    top = s.dump_top()
    assert_bits(top[1], g["top_left"], "fog scene BVH left links")
    assert_bits(top[2], g["top_right"], "fog scene BVH right links")
    assert_bits(top[3], g["top_leaf_ids"], "fog scene BVH leaf order")
    W, H, S, B = (int(v) for v in g["cfg"])
    out = s.render(orc.camera((0.0, 3.0, -10.0), (0.0, 0.0, 1.0)), W, H, S, B, int(g["seed"]), threads=8)
    assert_bits(out["primary_id"].astype(np.int8), g["primary_id"], "fog scene ids")
    assert_bits(out["primary_t"], g["primary_t"], "fog scene distances")
    assert_bits(out["radiance"], g["radiance"], "fog scene radiance")
    assert_bits(out["rgba"], g["rgba"], "fog scene frame")
    hit, rec = s.trace(g["rays_o"], g["rays_d"])
    assert_bits(hit, g["hit"], "fog scene traced ids")
    assert_bits(rec[:, :7], g["rec"], "fog scene traced records")


def test_portable_logf_is_the_rounded_double_log(orc):
    import math
    L = orc.lib()
    rng = np.random.default_rng(3)
    xs = np.concatenate([(rng.integers(1, 2 ** 31, 20000) / 2.0 ** 31).astype(np.float32), np.float32([1.0, 0.5, 2.0 ** -31, 3.0])])
    for x in xs:
        assert L.orc_pm_logf(float(x)) == np.float32(math.log(float(x)))
    assert L.orc_pm_logf(0.0) == -np.inf


def test_c3_scene_at_full_size(orc, golden, assets):
    """the 100 000-primitive scene: the oracle's tree is the reference's, and so are sample (0,0) of the 480 x 270 map
    and an 8 x 8 corner of the 256 spp crop (ids, distances, draw counts; colour through libm like the reference)"""
    g = golden("scene_c3")
    obj, tex, mat = synthetic_tables(int(g["n"]))
    s = build_table_scene(orc.Scene(orc.MATH_LIBM), tex, mat, obj, [[0.0, len(mat) - 1]], None, None, assets["sky"], int(g["build_key"]))
    for name, got in zip(("boxes", "left", "right", "leaf_ids"), s.dump_top()):
        assert_bits(got[::1000], g[f"top_{name}_rows"], f"C3 tree {name}, every 1000th row")
        assert hashlib.sha256(np.ascontiguousarray(got).tobytes()).digest() == g[f"top_{name}_sha256"].tobytes(), f"C3 tree {name}"
    cam = orc.camera(g["cam_pos"], g["cam_look"])
    Wm, Hm, Sm, Bm = (int(v) for v in g["map_cfg"])
    out = s.render(cam, Wm, Hm, Sm, Bm, int(g["seed"]), threads=8, want=("primary_id", "primary_t"))
    assert_bits(out["primary_id"].reshape(Hm, Wm), g["map_primary_id"], "C3 map ids")
    assert_bits(out["primary_t"].reshape(Hm, Wm), g["map_primary_t"], "C3 map distances")
    W, H, S, B = (int(v) for v in g["cfg"])
    x0, y0, x1, y1 = (int(v) for v in g["crop"])
    out = s.render(cam, W, H, S, B, int(g["seed"]), rect=(x0, y0, x0 + 8, y0 + 8), threads=8, want=("primary_id", "primary_t", "rng_draws", "rgba", "radiance"))
    sel = (np.arange(8)[:, None] * (x1 - x0) + np.arange(8)[None, :]).reshape(-1)
    assert_bits(out["primary_id"], g["crop_primary_id"][sel], "C3 crop ids")
    assert_bits(out["primary_t"], g["crop_primary_t"][sel], "C3 crop distances")
    assert_bits(out["rng_draws"].astype(np.uint16), g["crop_rng_draws"][sel], "C3 crop draws")
    assert_bits(out["rgba"], g["crop_rgba"][:8, :8], "C3 crop colour")
    sub = (np.arange(8)[:, None] * 16 + np.arange(8)[None, :]).reshape(-1)     # the stored radiance covers 16 x 16 pixels
    assert_bits(out["radiance"], g["crop_radiance"][sub], "C3 crop radiance")
