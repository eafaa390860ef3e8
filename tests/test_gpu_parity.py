"""GPU parity: the HIP engine, called through the C ABI, against
  (1) the golden vectors written by the reference's own C++ (tests/golden, oracle/gen_golden.py), and
  (2) the CPU oracle (oracle/rt_oracle.c) running the same portable sinf/atan2f as the kernels.

Bars: primary-hit object ids, hit distances, ray counts, RNG draw counts and per-sample radiance are compared BIT FOR BIT
with the oracle (portable math) — integer, index and float work alike.  Against the reference's golden frames, ids and
distances are bit-exact; 8-bit colour may differ where libm's sinf/atan2f and the portable versions round differently
(a skybox texel or a checker cell flips): tolerance stated in assert_frame_close.
"""
import os

import numpy as np
import pytest

from conftest import assert_bits, build_table_scene, random_tables, synthetic_tables

pytestmark = pytest.mark.gpu

CAM_POS, CAM_LOOK = (0.0, 3.0, -10.0), (0.0, 0.0, 1.0)


def assert_frame_close(got, ref, what):
    """8-bit frames: >= 99.9 % of bytes identical, no byte off by more than 2 on >= 99.99 % of bytes, PSNR >= 50 dB."""
    assert got.shape == ref.shape
    d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    same = float((d == 0).mean())
    within2 = float((d <= 2).mean())
    mse = float((d.astype(np.float64) ** 2).mean())
    psnr = 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
    assert same >= 0.999 and within2 >= 0.9999 and psnr >= 50.0, f"{what}: identical {same:.6f}, within2 {within2:.6f}, PSNR {psnr:.1f} dB"
    return same, psnr


@pytest.fixture(scope="module")
def gpu_default(rt, build_key):
    scene = rt.default_scene(build_key)
    return scene


def make_engine(rt, scene, W, H, S, B, cam=None, **kw):
    e = rt.Engine(rt.default_config(width=W, height=H, samples_per_axis=S, bounces=B, **kw))
    e.set_scene(scene)
    e.set_camera(cam if cam is not None else rt.camera_look(CAM_POS, CAM_LOOK))
    return e


def test_device_math_matches_oracle(rt, orc, gpu_default):
    e = make_engine(rt, gpu_default, 64, 64, 1, 1)
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-50, 50, 40000), rng.uniform(-1e5, 1e5, 20000), rng.uniform(-1e9, 1e9, 4000),
                        rng.normal(size=20000) * 1e-3, [0.0, -0.0, 1e-30, -1e-30, np.inf, -np.inf, np.nan, 3.14159274, 1.57079637, 1e38]]).astype(np.float32)
    y = rng.normal(size=len(x)).astype(np.float32)
    y[:50] = 0.0
    y[50:100] = -0.0
    x[100:150] = 0.0
    x[150:160] = -0.0
    y[150:155] = 0.0
    s, a = e.eval_math(x, y)
    L = orc.lib()
    s_ref = np.array([L.orc_pm_sinf(float(v)) for v in x], np.float32)
    a_ref = np.array([L.orc_pm_atan2f(float(b), float(c)) for b, c in zip(y, x)], np.float32)
    assert_bits(s, s_ref, "portable sinf")
    assert_bits(a, a_ref, "portable atan2f")
    # and they are good approximations of the functions the reference calls
    fin = np.isfinite(x) & (np.abs(x) < 1e5)
    assert np.max(np.abs(s[fin].astype(np.float64) - np.sin(x[fin].astype(np.float64)))) < 1e-7


def test_reciprocal_division_is_ieee_division(rt, gpu_default):
    """the slab test's a/b through the per-ray reciprocal must give the bits of a true division: explicit awkward pairs
    plus ~4e9 generated pairs whose quotients sit on or next to representable values and rounding midpoints"""
    e = make_engine(rt, gpu_default, 64, 64, 1, 1)
    rng = np.random.default_rng(17)
    n = 1 << 20
    b = (rng.choice([-1.0, 1.0], n) * np.exp2(rng.uniform(-19.9, 0.1, n))).astype(np.float32)
    a = (rng.normal(size=n) * np.exp2(rng.uniform(-30, 38, n))).astype(np.float32)
    a[:4096] = 0.0
    a[4096:8192] = -0.0
    a[8192:16384] = (b[8192:16384].astype(np.float64) * rng.integers(1, 1 << 24, 8192)).astype(np.float32)   # exact quotients
    a[16384:16640] = np.arange(256)            # texel bytes over 255 (World.cpp:192, DataUtility.cpp:120)
    b[16384:16640] = 255.0
    bad, first = e.check_division(a, b, seed=5, threads=1 << 22, rounds=256)
    assert bad == 0, f"{bad} quotients differ, first (a, b, got, want) = {first}"


def test_device_rng_matches_golden(rt, golden, gpu_default):
    e = make_engine(rt, gpu_default, 64, 64, 1, 1)
    g = golden("rng")
    keys = np.repeat(g["keys"], 16)
    ctr = np.tile(np.arange(16, dtype=np.uint32), len(g["keys"]))
    r, _ = e.eval_rng(keys, ctr)
    assert_bits(r.reshape(len(g["keys"]), 16), g["rand31"], "rand31")
    r2, ud = e.eval_rng(np.full(32, 0xDEADBEEF, np.uint32), np.arange(32, dtype=np.uint32))
    assert_bits(ud, g["random_unit_distance"], "RandomUnitDistance")


def test_trace_default_scene(rt, golden, oracle_default, gpu_default):
    g = golden("trace_default")
    e = make_engine(rt, gpu_default, 64, 64, 1, 1)
    hit, rec = e.trace(g["rays_o"], g["rays_d"])
    assert_bits(hit, g["hit"], "closest-hit object ids vs the reference")
    assert_bits(rec[:, 0], g["rec"][:, 0], "HitInfo.distance vs the reference (stale values included)")
    h2, r2 = oracle_default.trace(g["rays_o"], g["rays_d"])
    m = hit >= 0
    # point and normal bit for bit; uv of plane hits is never written by the reference
    assert_bits(rec[m, 1:7], r2[m, 1:7], "hit point and normal vs the oracle")
    tri = hit == 2
    assert_bits(rec[tri, 7:], r2[tri, 7:], "triangle uv vs the oracle")
    cube = hit == 0
    assert_bits(rec[cube, 7:], r2[cube, 7:], "cube uv vs the oracle")
    sph = m & ~tri & ~cube & (hit != 18)
    assert_bits(rec[sph, 7:], r2[sph, 7:], "sphere uv vs the oracle (portable atan2f)")
    # against the reference's libm atan2f the sphere uv may differ in the last place
    assert np.max(np.abs(rec[sph, 7:] - g["rec"][sph, 7:])) < 1e-6


def test_surface_rays_and_stale_hitinfo(rt, golden, oracle_default, gpu_default):
    """bounce-like rays from sphere surfaces, including the stale HitInfo.distance of the BVH's leftmost leaf"""
    g = golden("trace_default")
    e = make_engine(rt, gpu_default, 128, 128, 1, 3)
    hit, rec = e.trace(g["surf_o"], g["surf_d"])
    assert_bits(hit, g["surf_hit"], "surface-ray ids vs the reference")
    assert_bits(rec[:, 0], g["surf_rec"][:, 0], "surface-ray HitInfo.distance vs the reference")
    assert ((hit == -1) & (rec[:, 0] < 0)).sum() > 100
    m = hit >= 0
    assert_bits(rec[m, 1:7], g["surf_rec"][m, 1:7], "surface-ray hit point and normal vs the reference")
    col = e.ray_color(g["surf_o"], g["surf_d"], int(g["surf_color_bounces"]), g["surf_keys"])
    assert_bits(col, oracle_default.ray_color(g["surf_o"], g["surf_d"], int(g["surf_color_bounces"]), g["surf_keys"]), "GetRayColor vs the oracle")
    assert np.all(col.view(np.uint32) == g["surf_color"].view(np.uint32), axis=1).mean() > 0.995


def test_ray_color_default_scene(rt, golden, oracle_default, gpu_default):
    g = golden("trace_default")
    n = len(g["keys"])
    e = make_engine(rt, gpu_default, 256, 256, 1, 4)
    col = e.ray_color(g["rays_o"][:n], g["rays_d"][:n], 4, g["keys"])
    ref = oracle_default.ray_color(g["rays_o"][:n], g["rays_d"][:n], 4, g["keys"])
    assert_bits(col, ref, "GetRayColor vs the oracle")
    # vs the reference's own floats: equal except where a libm/portable rounding difference moved a texel or a checker cell
    same = np.all(col.view(np.uint32) == g["ray_color"].view(np.uint32), axis=1).mean()
    assert same > 0.995, same


def test_ray_queries_off_the_unit_sphere_stay_inside_the_tables(rt, oracle_default, gpu_default):
    """The skybox and texture lookups are not clamped in the reference (World.cpp:173-193, DataUtility.h:82): a direction
    that is not of unit length indexes rows past the image there.  The engine holds such an index in the table's padding
    row; rays with unit directions in the same call are untouched."""
    o = np.array([[0, 3, -10]] * 5, dtype=np.float32)
    d = np.array([[0, 5, 0], [0, -7, 0.1], [3, 4, 12], [-2, 9, 1], [0, 0, 1]], dtype=np.float32)
    keys = np.arange(5, dtype=np.uint32) * 2654435761 + 5
    e = make_engine(rt, gpu_default, 64, 64, 1, 3)
    col = e.ray_color(o, d, 3, keys)
    assert col.shape == (5, 3) and np.isfinite(col).all()
    assert_bits(col[4:], oracle_default.ray_color(o[4:], d[4:], 3, keys[4:]), "the unit-length ray of the call")


@pytest.mark.parametrize("W,H,S,B", [(256, 144, 3, 3), (160, 90, 2, 8), (67, 45, 4, 0)])
def test_render_small_frames_bit_exact(rt, oracle_default, gpu_default, W, H, S, B):
    cam = rt.camera_look(CAM_POS, CAM_LOOK)
    e = make_engine(rt, gpu_default, W, H, S, B, cam)
    e.keep_samples(True)
    frame = e.render_frame(11)
    got = e.read_samples()
    ref = oracle_default.render(cam, W, H, S, B, 11, threads=8, want=("radiance", "primary_id", "primary_t", "rng_draws", "rgba"))
    assert_bits(got["primary_id"], ref["primary_id"], "primary ids")
    assert_bits(got["primary_t"], ref["primary_t"], "primary distances")
    assert_bits(got["rng_draws"], ref["rng_draws"], "RNG draws per sample")
    assert_bits(got["radiance"], ref["radiance"], "per-sample radiance")
    assert_bits(frame, ref["rgba"], "8-bit frame")
    st = e.stats()
    assert st["paths"] == W * H * S * S


def test_c1_frame_vs_reference_golden(rt, golden, oracle_default, gpu_default):
    """The reference's own constants (1280x720, 9 spp, 3 bounces): every primary-hit id equals the reference's."""
    g = golden("frame_c1")
    W, H, S, B = (int(v) for v in g["cfg"])
    seed = int(g["seed"])
    e = make_engine(rt, gpu_default, W, H, S, B, max_batch_paths=1 << 30)
    e.keep_samples(True)
    frame = e.render_frame(seed)
    got = e.read_samples(want=("primary_id", "primary_t", "radiance", "rng_draws"))
    assert_bits(got["primary_id"].astype(np.int8), g["primary_id"], "C1 primary ids vs the reference")
    x0, y0, x1, y1 = (int(v) for v in g["crop"])
    sel = (np.arange(y0, y1)[:, None] * W + np.arange(x0, x1)[None, :]).reshape(-1)
    assert_bits(got["primary_t"][sel], g["crop_primary_t"], "C1 crop distances vs the reference")
    assert_bits(got["rng_draws"][sel], g["crop_rng_draws"], "C1 crop RNG draws vs the reference")
    same = np.all(got["radiance"][sel].view(np.uint32) == g["crop_radiance"].view(np.uint32), axis=2).mean()
    assert same > 0.995, f"crop radiance identical for {same:.4f} of samples"
    assert_frame_close(frame, g["rgba"], "C1 frame vs the reference")
    # and bit-exact against the oracle running the kernels' own sinf/atan2f, on a band of rows
    ref = oracle_default.render(rt.camera_look(CAM_POS, CAM_LOOK), W, H, S, B, seed, rect=(0, 300, W, 420), threads=8, want=("rgba", "radiance"))
    assert_bits(frame[300:420], ref["rgba"], "C1 rows 300..420 vs the oracle")
    band = (np.arange(300, 420)[:, None] * W + np.arange(W)[None, :]).reshape(-1)
    assert_bits(got["radiance"][band], ref["radiance"], "C1 band radiance vs the oracle")


def test_c2_crops_and_s0_ids(rt, golden, oracle_default, gpu_default):
    """1920x1080, 64 spp, 8 bounces: golden crops through the window API, and sample (0,0) ids of every pixel."""
    g = golden("frame_c2")
    W, H, S, B = (int(v) for v in g["cfg"])
    seed = int(g["seed"])
    cam = rt.camera_look(CAM_POS, CAM_LOOK)
    e = make_engine(rt, gpu_default, W, H, S, B)
    e.keep_samples(True)
    for i in range(3):
        rect = tuple(int(v) for v in g[f"crop{i}"])
        e.set_window(rect)
        e.render(seed)
        frame = e.read_frame()
        got = e.read_samples()
        assert_bits(got["primary_id"].astype(np.int8), g[f"crop{i}_primary_id"], f"C2 crop{i} ids vs the reference")
        assert_bits(got["primary_t"], g[f"crop{i}_primary_t"], f"C2 crop{i} distances vs the reference")
        assert_bits(got["rng_draws"].astype(np.uint16), g[f"crop{i}_rng_draws"], f"C2 crop{i} RNG draws vs the reference")
        x0, y0, x1, y1 = rect
        assert_frame_close(frame[y0:y1, x0:x1], g[f"crop{i}_rgba"], f"C2 crop{i} colour vs the reference")
        ref = oracle_default.render(cam, W, H, S, B, seed, rect=rect, threads=8, want=("radiance", "rgba"))
        assert_bits(got["radiance"], ref["radiance"], f"C2 crop{i} radiance vs the oracle")
        assert_bits(frame[y0:y1, x0:x1], ref["rgba"], f"C2 crop{i} colour vs the oracle")
    e.close()
    e1 = make_engine(rt, gpu_default, W, H, 1, B, max_batch_paths=1 << 30)
    e1.keep_samples(True)
    e1.render(seed)
    got = e1.read_samples(want=("primary_id", "primary_t"))
    assert_bits(got["primary_id"].astype(np.int8).reshape(H, W), g["s0_primary_id"], "C2 sample-0 ids vs the reference, every pixel")
    assert_bits(got["primary_t"].reshape(H, W)[::8], g["s0_primary_t_rows"], "C2 sample-0 distances vs the reference")


def test_c2_thumbnail(rt, golden, gpu_default):
    g = golden("frame_c2")
    W, H, S, B = (int(v) for v in g["thumb_cfg"])
    e = make_engine(rt, gpu_default, W, H, S, B)
    e.keep_samples(True)
    frame = e.render_frame(int(g["seed"]))
    got = e.read_samples(want=("primary_id",))
    assert_bits(got["primary_id"].astype(np.int8), g["thumb_primary_id"], "thumbnail ids vs the reference")
    assert_frame_close(frame, g["thumb_rgba"], "thumbnail colour vs the reference")


def test_custom_scene_all_materials(rt, orc, golden, assets):
    """every material and texture kind, Y/Z-rotated and moved mesh instances, inverted-box spheres, two planes"""
    g = golden("scene_custom")
    key = int(g["build_key"])
    args = (g["tex"], g["mat"], g["obj"], g["plane"], assets["verts"], assets["tex"], assets["sky"], key)
    scene = build_table_scene(rt.Scene(), *args)
    osc = build_table_scene(orc.Scene(orc.MATH_PORTABLE), *args)
    for a, b, n in zip(scene.dump_top(), osc.dump_top(), ("boxes", "left", "right", "leaf")):
        assert_bits(a, b, "custom BVH " + n)
    W, H, S, B = (int(v) for v in g["cfg"])
    cam = rt.camera_look(g["cam_pos"], g["cam_look"])
    e = make_engine(rt, scene, W, H, S, B, cam)
    e.keep_samples(True)
    frame = e.render_frame(int(g["seed"]))
    got = e.read_samples()
    assert_bits(got["primary_id"].astype(np.int8), g["primary_id"], "custom ids vs the reference")
    assert_bits(got["primary_t"], g["primary_t"], "custom distances vs the reference")
    assert_frame_close(frame, g["rgba"], "custom colour vs the reference")
    ref = osc.render(cam, W, H, S, B, int(g["seed"]), threads=8, want=("radiance", "rgba", "rng_draws"))
    assert_bits(got["radiance"], ref["radiance"], "custom radiance vs the oracle")
    assert_bits(got["rng_draws"], ref["rng_draws"], "custom RNG draws vs the oracle")
    assert_bits(frame, ref["rgba"], "custom frame vs the oracle")


def test_synthetic_scene(rt, orc, golden, assets):
    """a scaled-down C3: 3000 spheres/boxes from the PCG32 stream, same BVH as the reference builds"""
    g = golden("scene_synthetic")
    n, key = int(g["n"]), int(g["build_key"])
    obj, tex, mat = synthetic_tables(n)
    plane = [[0.0, len(mat) - 1]]
    scene = build_table_scene(rt.Scene(), tex, mat, obj, plane, None, None, assets["sky"], key)
    top = scene.dump_top()
    assert_bits(top[1], g["top_left"], "synthetic BVH left links vs the reference")
    assert_bits(top[2], g["top_right"], "synthetic BVH right links vs the reference")
    assert_bits(top[3], g["top_leaf_ids"], "synthetic BVH leaf order vs the reference")
    W, H, S, B = (int(v) for v in g["cfg"])
    cam = rt.camera_look((0.0, 20.0, -150.0), (0.0, 0.0, 1.0))
    e = make_engine(rt, scene, W, H, S, B, cam)
    e.keep_samples(True)
    frame = e.render_frame(int(g["seed"]))
    got = e.read_samples(want=("primary_id", "primary_t", "radiance"))
    assert_bits(got["primary_id"].astype(np.int16), g["primary_id"], "synthetic ids vs the reference")
    assert_bits(got["primary_t"].reshape(H, W, S * S)[::6], g["primary_t_rows"], "synthetic distances vs the reference")
    assert_frame_close(frame, g["rgba"], "synthetic colour vs the reference")
    osc = build_table_scene(orc.Scene(orc.MATH_PORTABLE), tex, mat, obj, plane, None, None, assets["sky"], key)
    ref = osc.render(cam, W, H, S, B, int(g["seed"]), threads=8, want=("radiance", "rgba"))
    assert_bits(got["radiance"], ref["radiance"], "synthetic radiance vs the oracle")
    assert_bits(frame, ref["rgba"], "synthetic frame vs the oracle")


def test_camera_path_last_frame(rt, golden, gpu_default):
    g = golden("camera_path")
    cam = rt.camera_look(CAM_POS, CAM_LOOK)
    for f in range(240):
        cam = rt.camera_input(cam, 3.0, 1.0, w=int(f < 120), d=int(f >= 120))
        assert_bits(cam, g["cams"][f], f"camera after frame {f}")
    W, H, S, B = (int(v) for v in g["last_cfg"])
    e = make_engine(rt, gpu_default, W, H, S, B, cam)
    e.keep_samples(True)
    frame = e.render_frame(7)
    got = e.read_samples(want=("primary_id",))
    assert_bits(got["primary_id"].astype(np.int8), g["last_primary_id"], "moved-camera ids vs the reference")
    assert_frame_close(frame, g["last_rgba"], "moved-camera colour vs the reference")


def test_batching_tiles_and_ranks_do_not_change_pixels(rt, gpu_default):
    """size-independent properties at a ragged size: batch size, tile size and the rank split leave every byte unchanged"""
    import torch
    W, H, S, B = 333, 187, 3, 5
    base = make_engine(rt, gpu_default, W, H, S, B).render_frame(3)
    small = make_engine(rt, gpu_default, W, H, S, B, max_batch_paths=50000).render_frame(3)
    assert_bits(small, base, "small batches")
    tiles = make_engine(rt, gpu_default, W, H, S, B, tile_w=16, tile_h=8, max_batch_paths=70000).render_frame(3)
    assert_bits(tiles, base, "16x8 tiles")
    # a frame smaller than one row of tiles, split in two: each rank's packed buffer (one padded 64 x 64 tile) is larger
    # than the frame itself
    tiny = make_engine(rt, gpu_default, 65, 33, 2, 2)
    tiny_base = tiny.render_frame(3)
    nb = tiny.tile_bytes(2)
    assert nb == 64 * 64 * 4 and 2 * nb > 65 * 33 * 4
    gathered = torch.zeros(2 * nb, dtype=torch.uint8, device="cuda")
    for r in range(2):
        tiny.render(3, rank=r, n_ranks=2, d_out=gathered.data_ptr() + r * nb)
    out = torch.zeros(33 * 65 * 4, dtype=torch.uint8, device="cuda")
    tiny.assemble(gathered.data_ptr(), 2, d_out=out.data_ptr())
    tiny.sync()
    assert_bits(out.cpu().numpy().reshape(33, 65, 4), tiny_base, "65 x 33 frame, 2-rank tile split")
    for n_ranks in (2, 3, 8):
        e = make_engine(rt, gpu_default, W, H, S, B, tile_w=32, tile_h=32)
        nb = e.tile_bytes(n_ranks)
        gathered = torch.zeros(n_ranks * nb, dtype=torch.uint8, device="cuda")
        for r in range(n_ranks):
            e.render(3, rank=r, n_ranks=n_ranks, d_out=gathered.data_ptr() + r * nb)
        out = torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda")
        e.assemble(gathered.data_ptr(), n_ranks, d_out=out.data_ptr())
        e.sync()
        assert_bits(out.cpu().numpy().reshape(H, W, 4), base, f"{n_ranks}-rank tile split")


def test_batching_tiles_and_ranks_on_a_scene_read_from_memory(rt, orc, assets):
    """the same properties where the scene is read from memory (four meshes and 61 spheres and cubes, no fog: the engine's own
    hierarchy, near side first, pair steps) and the camera looks along the z axis without a lens offset, so that the rays of
    one sample column have d.x = 0 exactly and are handed over to the walk through the reference's hierarchy, in whichever
    batch or share they fall: small batches, small tiles, windows and rank splits leave every byte unchanged, and the
    whole frame is the oracle's"""
    import torch
    tex, mat, obj, plane, (pos, look) = random_tables(9, fog=False)
    assert (obj[:, 0] == 3).sum() == 4 and len(obj) == 65
    args = (tex, mat, obj, plane, assets["verts"], assets["tex"], assets["sky"], 0xABCD)
    scene = build_table_scene(rt.Scene(), *args)
    osc = build_table_scene(orc.Scene(orc.MATH_PORTABLE), *args)
    assert scene.flat_info()["device_bytes"] > 160 * 1024 and scene.regroup_info()["has_regrouped_layout"] == 1
    W, H, S, B = 200, 120, 2, 4
    cam = rt.camera_look(pos, (0.0, 0.0, 1.0))
    kw = dict(dof_intensity=0.0)
    ref = osc.render(cam, W, H, S, B, 3, dof=0.0, threads=8, want=("rgba",))
    base = make_engine(rt, scene, W, H, S, B, cam, **kw).render_frame(3)
    assert_bits(base, ref["rgba"], "frame vs the oracle")
    assert_bits(make_engine(rt, scene, W, H, S, B, cam, max_batch_paths=9000, **kw).render_frame(3), base, "small batches")
    assert_bits(make_engine(rt, scene, W, H, S, B, cam, tile_w=16, tile_h=8, max_batch_paths=20000, **kw).render_frame(3), base, "16x8 tiles")
    e = make_engine(rt, scene, W, H, S, B, cam, **kw)
    e.set_window((37, 21, 150, 99))
    win = e.render_frame(3)
    assert_bits(win[21:99, 37:150], base[21:99, 37:150], "window vs frame")
    e.set_window(None)
    for n_ranks in (2, 5):
        e = make_engine(rt, scene, W, H, S, B, cam, tile_w=32, tile_h=32, max_batch_paths=30000, **kw)
        nb = e.tile_bytes(n_ranks)
        gathered = torch.zeros(n_ranks * nb, dtype=torch.uint8, device="cuda")
        for r in range(n_ranks):
            e.render(3, rank=r, n_ranks=n_ranks, d_out=gathered.data_ptr() + r * nb)
        out = torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda")
        e.assemble(gathered.data_ptr(), n_ranks, d_out=out.data_ptr())
        e.sync()
        assert_bits(out.cpu().numpy().reshape(H, W, 4), base, f"{n_ranks}-rank tile split")


def test_c2_full_frame_properties(rt, gpu_default, monkeypatch):
    """BASELINE's full size (1920x1080, 64 spp, 8 bounces): deterministic, alpha constant, windows agree with the frame"""
    W, H, S, B = 1920, 1080, 8, 8
    e = make_engine(rt, gpu_default, W, H, S, B)
    a = e.render_frame(7)
    st = e.stats()
    b = e.render_frame(7)
    assert_bits(a, b, "two renders of the same frame")
    assert (a[..., 0] == 255).all()
    assert st["paths"] == W * H * S * S and st["rays"] > st["paths"]
    assert st["rays_at_depth"][0] == st["paths"] and all(x >= y for x, y in zip(st["rays_at_depth"][:9], st["rays_at_depth"][1:10]))
    e2 = make_engine(rt, gpu_default, W, H, S, B)
    e2.set_window((928, 508, 992, 540))
    c = e2.render_frame(7)
    assert_bits(c[508:540, 928:992], a[508:540, 928:992], "window vs full frame")
    assert not c[:508].any()
    # a batch of this size runs its first bounce passes with 768-thread blocks; the same frame with 512-thread blocks
    monkeypatch.setenv("RT_WIDE_BLOCKS", "0")
    e3 = make_engine(rt, gpu_default, W, H, S, B)
    monkeypatch.delenv("RT_WIDE_BLOCKS")
    assert_bits(e3.render_frame(7), a, "512-thread against 768-thread bounce passes")
    # the frame in four batches, each still large enough for the 768-thread variants (whose camera-ray pass parks rays in
    # the queue of the pass after it): batches follow one another through the same queues
    e4 = make_engine(rt, gpu_default, W, H, S, B, max_batch_paths=1 << 25)
    assert_bits(e4.render_frame(7), a, "four batches against one")


def test_error_behaviour(rt, gpu_default):
    E = rt.engine
    with pytest.raises(rt.RtError):
        rt.Engine(rt.default_config(width=0))
    e = rt.Engine(rt.default_config(width=32, height=32))
    with pytest.raises(rt.RtError):
        e.render(1)                      # no scene, no camera
    e.set_scene(gpu_default)
    e.set_camera(rt.camera_look(CAM_POS, CAM_LOOK))
    with pytest.raises(rt.RtError):
        e.render(1, rank=2, n_ranks=2)
    with pytest.raises(rt.RtError, match="d_rgba"):
        e.render(1, rank=0, n_ranks=2)   # a rank's padded tiles can be larger than the engine's own frame buffer
    with pytest.raises(rt.RtError):
        e.set_window((0, 0, 33, 10))
    with pytest.raises(rt.RtError):
        e.read_samples()                 # not armed
    assert E.lib().rt_last_error()


def test_cancel_and_host_frame(rt, gpu_default):
    """FrameManager's Quit and presentation handshakes (FrameManager.cpp:31-38, 60-76) on the GPU: a render in flight can
    be abandoned without waiting for it, the next frame is untouched by that, and the page-locked host frame holds the
    bytes of the device frame — the whole frame after the render, its finished batches while the render goes on."""
    import time
    W, H, S, B = 1920, 1080, 16, 8          # 256 spp in 16 batches of 2^25 paths: about a hundred milliseconds
    e = make_engine(rt, gpu_default, W, H, S, B, max_batch_paths=1 << 25)
    e.render(21)
    e.sync()
    t0 = time.perf_counter()
    e.render(21)
    e.sync()
    t_full = time.perf_counter() - t0
    want = e.read_frame()
    assert_bits(e.map_frame(wait=True), want, "page-locked frame vs read_frame")
    assert want.any()
    # abandon a frame after a third of its time
    e.clear_frame()
    t0 = time.perf_counter()
    e.render(22)
    time.sleep(0.33 * t_full)
    snap = e.map_frame(wait=False).copy()      # while the render goes on
    e.cancel()
    e.sync()
    t_cancelled = time.perf_counter() - t0
    assert t_cancelled < 0.75 * t_full, f"a cancelled frame took {t_cancelled * 1e3:.0f} ms of {t_full * 1e3:.0f} ms"
    part = e.map_frame(wait=True).copy()
    done = part[..., 0] == 255                 # alpha is 255 in every rendered pixel, 0 in a cleared one
    assert 0 < done.mean() < 1, f"a cancelled frame should be partly rendered, {done.mean():.3f} is"
    assert (snap[..., 0] == 255).mean() <= done.mean()
    # the next frame does not know about it
    e.render(21)
    assert_bits(e.map_frame(wait=True), want, "frame after a cancelled one")
    # cancelling with nothing in flight, and a cancel that arrives after its frame has finished, change nothing
    e.cancel()
    e.render(21)
    e.sync()
    e.cancel()
    assert_bits(e.read_frame(), want, "frame after stray cancels")
    st = e.stats()
    assert st["paths"] == W * H * S * S
    print(f"full frame {t_full * 1e3:.0f} ms, cancelled after {0.33 * t_full * 1e3:.0f} ms: back in {t_cancelled * 1e3:.0f} ms, {done.mean():.2f} of the pixels done")


@pytest.mark.parametrize("ids", [[0], [0, 0], [0, 0, 0, 0, 0]])
def test_several_devices_behind_one_handle(rt, gpu_default, ids):
    """rt_multi: one process, one engine per listed device, tiles interleaved over them, peer copies to the first one.
    A one-GPU box lists its device several times, which runs the same code.  Every byte equals the one-engine frame."""
    W, H, S, B = 333, 187, 3, 5
    cam = rt.camera_look(CAM_POS, CAM_LOOK)
    base = make_engine(rt, gpu_default, W, H, S, B, cam).render_frame(3)
    m = rt.MultiEngine(rt.default_config(width=W, height=H, samples_per_axis=S, bounces=B, tile_w=32, tile_h=32), ids)
    assert len(m) == len(ids)
    m.set_scene(gpu_default)
    m.set_camera(cam)
    m.clear_frame()
    m.render(3)
    assert_bits(m.map_frame(wait=True), base, f"{len(ids)} engines behind one handle")
    assert m.poll() is True
    assert sum(m.stats(k)["paths"] for k in range(len(ids))) == W * H * S * S
    m.render(4)
    m.cancel()
    m.sync()
    m.render(3)
    assert_bits(m.map_frame(wait=True), base, "frame after a cancelled one")


def test_frames_queued_back_to_back_on_several_devices(rt, gpu_default):
    """rt_multi_render three times with no wait in between, a different seed each time: a device's next render zeroes and
    refills the packed tiles the display device still has to copy for the frame before, so it must wait for that copy on
    the GPU.  Large enough that every render is still running when the next one is queued; the last frame is checked."""
    W, H, S, B = 1920, 1080, 4, 6
    cam = rt.camera_look(CAM_POS, CAM_LOOK)
    one = make_engine(rt, gpu_default, W, H, S, B, cam)
    want = {seed: one.render_frame(seed) for seed in (31, 32, 33)}
    one.close()
    assert (want[32] != want[33]).any()
    m = rt.MultiEngine(rt.default_config(width=W, height=H, samples_per_axis=S, bounces=B), [0, 0, 0])
    m.set_scene(gpu_default)
    m.set_camera(cam)
    for seed in (31, 32, 33):
        m.render(seed)
    assert_bits(m.map_frame(wait=True), want[33], "third of three frames queued back to back")
    for seed in (33, 31, 32):                      # and once more, starting while nothing is in flight
        m.render(seed)
    assert_bits(m.map_frame(wait=True), want[32], "sixth frame")


def test_tile_renderer_frame_is_ordered_for_the_callers_stream(rt, gpu_default):
    """TileRenderer renders on a stream of its own; a copy the caller queues on the current stream right after
    render_frame (no synchronize in between) must see the finished frame"""
    import torch
    from raytracer_amd.distributed import TileRenderer
    W, H, S, B = 1920, 1080, 4, 6                  # several milliseconds of rendering
    e = make_engine(rt, gpu_default, W, H, S, B)
    want = e.render_frame(41)
    tr = TileRenderer(e, "cuda:0")
    host = torch.empty(H * W * 4, dtype=torch.uint8, pin_memory=True)
    for seed in (40, 41):
        frame = tr.render_frame(seed)
        host.copy_(frame, non_blocking=True)       # on the current stream, straight behind the call
    torch.cuda.current_stream().synchronize()
    assert_bits(host.numpy().reshape(H, W, 4), want, "frame copied on the caller's stream without a device-wide wait")


def test_host_frame_before_any_render_and_single_pixels(rt, gpu_default):
    """FrameManager hands out its zeroed texturePixels before the first frame has started (FrameManager.cpp:13,60-71); a
    caller of World::CalculateColorForScreenPosition gets a pixel's bytes without the frame around it"""
    W, H, S, B = 320, 180, 2, 3
    e = make_engine(rt, gpu_default, W, H, S, B)
    assert not e.map_frame(wait=True).any(), "the frame before any render must be zero"
    frame = e.render_frame(5)
    assert_bits(e.read_rect(17, 40, 18, 41), frame[40:41, 17:18], "one pixel")
    assert_bits(e.read_rect(300, 100, 320, 180), frame[100:180, 300:320], "a rectangle up to the frame's corner")
    e.set_window((17, 40, 18, 41))                 # the adapter's CalculateColorForScreenPosition
    e.clear_frame()
    e.render(5)
    assert_bits(e.read_rect(17, 40, 18, 41), frame[40:41, 17:18], "one pixel rendered on its own")
    with pytest.raises(rt.RtError):
        e.read_rect(0, 0, W + 1, 1)
    with pytest.raises(rt.RtError):                # 64-pixel batches of a 4K frame: more pass counters than an engine reserves
        rt.Engine(rt.default_config(width=3840, height=2160, samples_per_axis=1, bounces=1, max_batch_paths=64))


def test_front_end_adapter_matches_the_c_abi(rt, gpu_default, tmp_path):
    """raytracer_amd/compat_demo drives the FrameManager / World / Camera adapter (raytracer_amd/csrc/compat, the
    reference's class interface over the C ABI) like the reference's main loop, at the resolution and quality of the
    front end's Constants.h it was compiled against; its frame must be the frame the C ABI renders for the same
    scene key, frame seed and camera."""
    import os
    import subprocess
    from conftest import ROOT
    demo = os.path.join(ROOT, "raytracer_amd", "compat_demo")
    assert os.path.exists(demo), "raytracer_amd/compat_demo is not built (make -C raytracer_amd/csrc compat)"
    out = tmp_path / "frame.raw"
    env = dict(os.environ, RT_ASSET_DIR=os.path.join(ROOT, "assets"))
    log = subprocess.run([demo, "1", str(out)], env=env, capture_output=True, text=True, timeout=110)
    assert log.returncode == 0, log.stdout + log.stderr
    assert "Frame calculating time" in log.stdout          # LOG_BENCHMARK print of FrameManager::FinishFrame
    raw = np.fromfile(out, np.uint8)
    # the same front end with the frame split over "two devices" (this GPU listed twice): the same bytes
    out2 = tmp_path / "frame2.raw"
    log2 = subprocess.run([demo, "1", str(out2)], env=dict(env, RT_DEVICES="0,0"), capture_output=True, text=True, timeout=110)
    assert log2.returncode == 0, log2.stdout + log2.stderr
    assert_bits(np.fromfile(out2, np.uint8), raw, "adapter frame with RT_DEVICES=0,0")
    cfg = rt.default_config()                              # the values of Constants.h
    W, H = cfg.width, cfg.height
    assert raw.size == W * H * 4
    e = rt.Engine(cfg)
    e.set_scene(rt.default_scene(1))                       # World::buildKey of the adapter
    e.set_camera(rt.camera_look(CAM_POS, CAM_LOOK))
    assert_bits(raw.reshape(H, W, 4), e.render_frame(2), "adapter frame vs C ABI frame")   # first frame: frameSeed 2
    assert e.poll() is True


def _grazing_rays(boxes, n_origins, seed):
    """rays aimed at the corners and edge midpoints of the given boxes (float rounding puts them a hair inside or
    outside): the cases where a slab test's entry and exit distances nearly coincide"""
    rng = np.random.default_rng(seed)
    lo, hi = boxes[:, :3].astype(np.float64), boxes[:, 3:].astype(np.float64)
    pts = []
    for cx in (0.0, 0.5, 1.0):
        for cy in (0.0, 0.5, 1.0):
            for cz in (0.0, 0.5, 1.0):
                if (cx == 0.5) + (cy == 0.5) + (cz == 0.5) <= 1:          # corners and edge midpoints
                    pts.append(lo + (hi - lo) * np.array([cx, cy, cz]))
    pts = np.concatenate(pts)
    pts = pts[np.isfinite(pts).all(axis=1)]
    o = rng.uniform(-60.0, 60.0, (n_origins, 1, 3)) * np.array([1.0, 0.4, 1.0]) + np.array([0.0, 25.0, 0.0])
    d = pts[None, :, :] - o
    d /= np.linalg.norm(d, axis=2, keepdims=True)
    o = np.broadcast_to(o, d.shape)
    return o.reshape(-1, 3).astype(np.float32), d.reshape(-1, 3).astype(np.float32)


def test_grazing_rays_decide_like_the_oracle(rt, orc, golden, assets, oracle_default, gpu_default):
    """The box step decides on approximate quotients and falls back to exactly rounded ones when the decision is close;
    rays through box corners and along box edges are where that matters.  Both variants of the traversal loop: scene in
    LDS (ray_color on the default scene) and scene in global memory (trace; the 3000-primitive scene)."""
    e = make_engine(rt, gpu_default, 256, 256, 2, 3)    # ray_color needs queue room for every ray and 3 bounce levels
    o, d = _grazing_rays(gpu_default.dump_top()[0], 40, 1)
    hit, rec = e.trace(o, d)
    h2, r2 = oracle_default.trace(o, d)
    assert_bits(hit, h2, "default scene, grazing rays: ids")
    assert_bits(rec[:, 0], r2[:, 0], "default scene, grazing rays: distances")
    keys = np.arange(len(o), dtype=np.uint32) * 2654435761 + 17
    assert_bits(e.ray_color(o, d, 3, keys), oracle_default.ray_color(o, d, 3, keys), "default scene, grazing rays: radiance (LDS variant)")

    n, key = int(golden("scene_synthetic")["n"]), int(golden("scene_synthetic")["build_key"])
    obj, tex, mat = synthetic_tables(n)
    plane = [[0.0, len(mat) - 1]]
    scene = build_table_scene(rt.Scene(), tex, mat, obj, plane, None, None, assets["sky"], key)
    osc = build_table_scene(orc.Scene(orc.MATH_PORTABLE), tex, mat, obj, plane, None, None, assets["sky"], key)
    boxes = scene.dump_top()[0]
    o, d = _grazing_rays(boxes[:: max(1, len(boxes) // 400)], 6, 2)
    e2 = make_engine(rt, scene, 64, 64, 1, 1)
    hit, rec = e2.trace(o, d)
    h2, r2 = osc.trace(o, d)
    assert_bits(hit, h2, "synthetic scene, grazing rays: ids")
    assert_bits(rec[:, 0], r2[:, 0], "synthetic scene, grazing rays: distances")


def test_fog_scene(rt, orc, golden, assets):
    """Fog volumes (plain, X/Y/Z-rotated, moved) with the reference's alternative scene around them.  Fog draws random
    numbers inside the traversal: the engine notes leaves and tests them later, and must still test them in the
    reference's order, duplicates included.  Bit-exact against the oracle (portable logf); against the reference
    (libm logf) a free path may round differently, so ids get the same tolerance as colour."""
    g = golden("scene_fog")
    args = (g["tex"], g["mat"], g["obj"], g["plane"], None, None, assets["sky"], int(g["build_key"]))
    scene = build_table_scene(rt.Scene(), *args)
    osc = build_table_scene(orc.Scene(orc.MATH_PORTABLE), *args)
    W, H, S, B = (int(v) for v in g["cfg"])
    cam = rt.camera_look(CAM_POS, CAM_LOOK)
    e = make_engine(rt, scene, W, H, S, B, cam)
    e.keep_samples(True)
    frame = e.render_frame(int(g["seed"]))
    got = e.read_samples()
    ref = osc.render(cam, W, H, S, B, int(g["seed"]), threads=8, want=("radiance", "rgba", "rng_draws", "primary_id", "primary_t"))
    assert_bits(got["primary_id"], ref["primary_id"], "fog scene ids vs the oracle")
    assert_bits(got["primary_t"], ref["primary_t"], "fog scene distances vs the oracle")
    assert_bits(got["rng_draws"], ref["rng_draws"], "fog scene RNG draws vs the oracle")
    assert_bits(got["radiance"], ref["radiance"], "fog scene radiance vs the oracle")
    assert_bits(frame, ref["rgba"], "fog scene frame vs the oracle")
    same_ids = float((got["primary_id"].astype(np.int8) == g["primary_id"]).mean())
    assert same_ids >= 0.9999, f"fog scene ids vs the reference: {same_ids:.6f} identical"
    assert_frame_close(frame, g["rgba"], "fog scene colour vs the reference")
    hit, rec = e.trace(g["rays_o"], g["rays_d"])
    h2, r2 = osc.trace(g["rays_o"], g["rays_d"])
    assert_bits(hit, h2, "fog scene traced ids vs the oracle")
    assert_bits(rec[:, :7], r2[:, :7], "fog scene traced records vs the oracle")
    # batching must not change a fog scene either: the draw counter travels with the hit record
    e2 = make_engine(rt, scene, W, H, S, B, cam, max_batch_paths=W * S * S * 16)
    assert_bits(e2.render_frame(int(g["seed"])), frame, "fog scene, batched")


def test_rays_the_regrouped_hierarchy_is_not_proven_for(rt, oracle_default, gpu_default):
    """The engine walks a hierarchy of its own above the lowest nodes of a mesh's tree; that is exact for every ray whose
    slab-test quotients are numbers.  Rays with a direction component that is exactly zero (+0 or -0: the reference's
    test takes -0 for a direction that points the other way) — in the scene's frame, or only in the frame of the tree's
    rotated instance — are handed over to a walk through the reference's own hierarchy.  Closest hits and radiance of
    such rays through the tree's mesh, against the oracle and against an engine that walks the reference's hierarchy
    throughout, bit for bit; and the hand-over really happens."""
    rng = np.random.default_rng(5)
    n = 6000
    # origins around the tree (centre (0,5,0), 10 units high), directions along the axes and in the coordinate planes
    o = rng.uniform([-6, 0.2, -6], [6, 11, 6], (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    zero = rng.integers(0, 3, n)
    d[np.arange(n), zero] = np.where(rng.random(n) < 0.5, np.float32(0.0), np.float32(-0.0))
    second = rng.random(n) < 0.4                       # two zero components: axis-parallel rays
    d[second, (zero[second] + 1) % 3] = np.where(rng.random(second.sum()) < 0.5, np.float32(0.0), np.float32(-0.0))
    d /= np.sqrt((d.astype(np.float64) ** 2).sum(axis=1, keepdims=True)).astype(np.float32)
    # aimed at the tree from outside its box
    o[: n // 2] = (np.array([0, 5, 0], np.float32) - d[: n // 2] * rng.uniform(6, 15, (n // 2, 1)).astype(np.float32)
                   + rng.uniform(-3, 3, (n // 2, 3)).astype(np.float32) * (d[: n // 2] == 0))
    # a zero component only after the instance's rotation by 90 degrees about x (sin = 1, cos = -4.37e-8):
    # d'.y = cos * d.y - sin * d.z is exactly 0 for d.z = cos * d.y
    m = 500
    o2 = rng.uniform([-5, 1, -9], [5, 9, -7], (m, 3)).astype(np.float32)
    dy = rng.uniform(0.2, 1.0, m).astype(np.float32)
    cos_t = np.float32(np.cos(np.float32(90.0 * 0.01745329251994329576923690768489)))
    d2 = np.stack([rng.uniform(-0.3, 0.3, m).astype(np.float32), dy, cos_t * dy], axis=1).astype(np.float32)
    assert np.all(cos_t * d2[:, 1] - np.float32(1.0) * d2[:, 2] == 0)
    o = np.concatenate([o, o2])
    d = np.concatenate([d, d2])
    keys = (np.arange(len(o), dtype=np.uint64) * 2654435761 % (1 << 32)).astype(np.uint32)
    e = make_engine(rt, gpu_default, 128, 128, 1, 4)
    e0 = make_engine(rt, gpu_default, 128, 128, 1, 4, flags=rt.engine.FLAG_NO_REGROUP)
    assert gpu_default.regroup_info()["has_regrouped_layout"] == 1
    hit, rec = e.trace(o, d)
    hit0, rec0 = e0.trace(o, d)
    h_or, r_or = oracle_default.trace(o, d)
    assert_bits(hit, h_or, "ids of rays with zero direction components vs the oracle")
    assert_bits(rec[:, 0], r_or[:, 0], "their distances vs the oracle")
    assert_bits(hit, hit0, "ids vs the engine on the reference's hierarchy")
    assert_bits(rec, rec0, "hit records vs the engine on the reference's hierarchy")
    assert (hit == 2).sum() > 300                      # many of them do meet the tree
    col = e.ray_color(o, d, 4, keys)
    assert_bits(col, oracle_default.ray_color(o, d, 4, keys), "radiance of such rays vs the oracle")
    assert_bits(col, e0.ray_color(o, d, 4, keys), "radiance vs the engine on the reference's hierarchy")
    # the same through the kernels that read the scene from memory (near side first: pair steps, one box per step)
    for flags in (rt.engine.FLAG_NO_LDS, rt.engine.FLAG_NO_LDS | rt.engine.FLAG_NO_PAIRS, rt.engine.FLAG_NO_LDS | rt.engine.FLAG_NO_ORDER, rt.engine.FLAG_NO_WIDE):
        em = make_engine(rt, gpu_default, 128, 128, 1, 4, flags=flags)
        assert_bits(em.ray_color(o, d, 4, keys), col, f"radiance of such rays, scene read from memory (flags {flags})")
        em.close()
    # a frame through both hierarchies, and the work the regrouped one saves
    a, a0 = e.render_frame(3), e0.render_frame(3)
    assert_bits(a, a0, "frame through the regrouped and through the reference's hierarchy")
    assert e.stats()["box_tests"] < 0.95 * e0.stats()["box_tests"]
    # camera rays of that kind: without a lens offset (DEPTH_OF_FIELD_INTENSITY 0) the first sample column of the pixel
    # column at x = W / 2 looks along the camera's forward axis with d.x = 0 exactly — from in front of the tree, whose
    # trunk they meet, those rays are handed over in the camera rays' pass
    W, H, S, B = 64, 48, 2, 3
    cam = rt.camera_look((0.0, 4.0, -12.0), (0.0, 0.0, 1.0))
    p = make_engine(rt, gpu_default, W, H, S, B, cam, dof_intensity=0.0)
    p.keep_samples(True)
    frame = p.render_frame(11)
    got = p.read_samples()
    ref = oracle_default.render(cam, W, H, S, B, 11, dof=0.0, threads=4, want=("radiance", "rgba", "rng_draws", "primary_id", "primary_t"))
    for name in ("primary_id", "primary_t", "rng_draws", "radiance"):
        assert_bits(got[name], ref[name], f"frame without lens offsets: {name}")
    assert_bits(frame, ref["rgba"], "frame without lens offsets")
    assert (got["primary_id"].reshape(H, W, S * S)[:, W // 2, 0] == 2).sum() > 5      # (sample 0 of that column: the trunk)
    pm = make_engine(rt, gpu_default, W, H, S, B, cam, dof_intensity=0.0, flags=rt.engine.FLAG_NO_LDS)
    assert_bits(pm.render_frame(11), ref["rgba"], "frame without lens offsets, scene read from memory")
    pm.close()


def test_rays_outside_the_fast_paths(rt, orc, assets):
    """Rays the shortcuts are not proven for must take the reference's arithmetic as written: origins far outside the
    scene (beyond the distance up to which a mesh instance's entry box is proven, and beyond the range in which the
    reciprocal path reproduces division), direction components below 2^-20 or above 2^20 in magnitude, directions that
    are not of unit length by many orders of magnitude, infinities and NaNs.  Closest hits and radiance against the
    oracle bit for bit, with the scene on chip and read from memory (near side first, pair steps), regrouped or not.
    The tree's mesh carries a plain colour here: with its image texture the reference looks the cut-out up outside the
    image for such rays (texture coordinates computed from a hit point that is off by many units, DataUtility.h:82)."""
    tex = np.array([[0, 0.8, 0.7, 0.6, 0, 0, 0], [1, 0.9, 0.9, 0.9, 0.2, 0.5, 0.3]], np.float32)
    mat = np.array([[1, 0, 0, 0, 0], [2, 1, 0, 0, 0]], np.float32)
    obj = [[3, 10.0, 0, 90.0, 0.0, 5.0, 0.0, 1, 0, 0], [3, 6.0, 1, 30.0, 7.0, 3.0, 5.0, 1, 0, 0]]
    r = np.random.default_rng(12)
    for _ in range(9):
        obj.append([0, *r.uniform([-9, 0.5, -6], [9, 8, 12]), r.uniform(0.5, 2.0), 0, 0, 0, 0, int(r.integers(0, 2))])
    obj = np.array(obj, np.float32)
    args = (tex, mat, obj, [[0.0, 1]], assets["verts"], assets["tex"], assets["sky"], 31337)
    scene = build_table_scene(rt.Scene(), *args)
    osc = build_table_scene(orc.Scene(orc.MATH_PORTABLE), *args)
    assert scene.regroup_info()["has_regrouped_layout"] == 1
    rng = np.random.default_rng(77)
    n = 4000
    target = np.array([0, 5, 0], np.float32) + rng.uniform(-3, 3, (n, 3)).astype(np.float32)      # around the first tree
    u = rng.normal(size=(n, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    dist = np.float32(10.0) ** rng.uniform(1, 13, (n, 1)).astype(np.float32)                      # 10 .. 1e13 units away
    o = (target - u * dist).astype(np.float32)
    d = u.astype(np.float32)
    # a share with one tiny component, a share scaled far off unit length, a few with infinities and NaNs
    k = n // 4
    d[:k, rng.integers(0, 3, k)] *= np.float32(1e-9)
    d[k:2 * k] *= (np.float32(10.0) ** rng.uniform(-12, 12, (k, 1))).astype(np.float32)
    o[2 * k:2 * k + 40] = rng.uniform(-8, 8, (40, 3)).astype(np.float32)
    d[2 * k:2 * k + 10, 0] = np.inf
    d[2 * k + 10:2 * k + 20, 1] = -np.inf
    d[2 * k + 20:2 * k + 30, 2] = np.nan
    o[2 * k + 30:2 * k + 40, 1] = np.inf
    # near origins with ordinary directions beside them, so that waves hold both kinds
    o[3 * k:] = rng.uniform([-9, 0.5, -12], [9, 10, 9], (n - 3 * k, 3)).astype(np.float32)
    keys = (np.arange(n, dtype=np.uint64) * 2246822519 % (1 << 32)).astype(np.uint32)
    h_or, r_or = osc.trace(o, d)
    # (radiance only for rays whose direction is of unit length: the reference's sky lookup is not clamped and indexes
    # outside its image for any other, World.cpp:173-193)
    fin = np.isfinite(o).all(axis=1) & np.isfinite(d).all(axis=1)
    fin[k:2 * k] = False
    c_or = osc.ray_color(o[fin], d[fin], 3, keys[fin])
    assert (h_or >= 0).sum() > 500 and (h_or[:3 * k] == 0).sum() > 50        # far rays do meet the tree
    cam = rt.camera_look(CAM_POS, CAM_LOOK)
    E = rt.engine
    for flags in (0, E.FLAG_NO_WIDE, E.FLAG_NO_REGROUP, E.FLAG_NO_LDS, E.FLAG_NO_LDS | E.FLAG_NO_PAIRS, E.FLAG_NO_LDS | E.FLAG_NO_REGROUP, E.FLAG_NO_LDS | E.FLAG_NO_ORDER):
        e = make_engine(rt, scene, 64, 64, 1, 3, cam, flags=flags)
        hit, rec = e.trace(o, d)
        assert_bits(hit, h_or, f"ids of rays outside the fast paths (flags {flags})")
        assert_bits(rec[:, 0], r_or[:, 0], f"their distances (flags {flags})")
        assert_bits(e.ray_color(o[fin], d[fin], 3, keys[fin]), c_or, f"their radiance (flags {flags})")
        e.ray_color(o[~fin], d[~fin], 3, keys[~fin])          # the others: no fault, whatever the colour
        e.close()


@pytest.mark.parametrize("case", ["single_sphere_no_plane", "single_fog", "one_pixel", "no_bounces", "max_bounces", "two_objects", "three_objects", "five_objects"])
def test_degenerate_scenes_and_sizes(rt, orc, assets, case):
    """the small ends of the input space: a BVH whose root has the same object as both children (Object.cpp:198-200), a
    scene without planes, a 1 x 1 frame, zero bounces, the largest bounce count the engine accepts; trees of two, three
    and five objects (a root whose children are leaves; a root with one leaf and one node; the smallest tree in which a
    node's two children both have boxes) — each also through the kernels that read the scene from memory, whose walk
    starts at the root's children"""
    tex = [[0, .7, .6, .5, 0, 0, 0], [1, 1, 1, 1, .1, .4, .2]]
    mat = [[1, 0, 0, 0, 0], [2, 1, 0, 0, 0], [3, 0, 0, 0, 0]]
    obj = np.zeros((3, 10), np.float32)
    obj[0] = [0, 0.5, 2.5, 2.0, 2.0, 0, 0, 0, 0, 0]
    obj[1] = [1, -4, 0, 1, -2, 3, 4, 0, 0, 0]
    obj[2] = [4, 2.5, 1.5, -1, 1.5, 0.7, 1, 25.0, 0, 2]
    plane = [[0.0, 1]]
    W, H, S, B = 96, 54, 2, 4
    if case == "single_sphere_no_plane":
        obj, plane = obj[:1], np.zeros((0, 2), np.float32)
    elif case == "single_fog":
        obj = obj[2:]
    elif case == "one_pixel":
        W, H = 1, 1
    elif case == "no_bounces":
        B = 0
    elif case == "max_bounces":
        B, W, H = 32, 48, 27
    elif case in ("two_objects", "three_objects", "five_objects"):
        more = np.zeros((5, 10), np.float32)
        more[0] = obj[0]
        more[1] = obj[1]
        more[2] = [0, -1.5, 1.0, 4.0, 1.0, 0, 0, 0, 0, 1]
        more[3] = [1, 3.5, 0.0, 5.0, 5.0, 2.5, 7.0, 0, 0, 0]
        more[4] = [0, 0.0, 6.0, 9.0, 3.0, 0, 0, 0, 0, 1]
        obj = more[:{"two_objects": 2, "three_objects": 3, "five_objects": 5}[case]]
    args = (tex, mat, obj, plane, None, None, assets["sky"], 77)
    scene = build_table_scene(rt.Scene(), *args)
    osc = build_table_scene(orc.Scene(orc.MATH_PORTABLE), *args)
    cam = rt.camera_look(CAM_POS, CAM_LOOK)
    e = make_engine(rt, scene, W, H, S, B, cam)
    e.keep_samples(True)
    frame = e.render_frame(9)
    got = e.read_samples()
    ref = osc.render(cam, W, H, S, B, 9, threads=4, want=("radiance", "rgba", "rng_draws", "primary_id", "primary_t"))
    assert_bits(got["primary_id"], ref["primary_id"], case + ": ids")
    assert_bits(got["primary_t"], ref["primary_t"], case + ": distances")
    assert_bits(got["rng_draws"], ref["rng_draws"], case + ": RNG draws")
    assert_bits(got["radiance"], ref["radiance"], case + ": radiance")
    assert_bits(frame, ref["rgba"], case + ": frame")
    E = rt.engine
    for flags in (E.FLAG_NO_LDS, E.FLAG_NO_LDS | E.FLAG_NO_PAIRS, E.FLAG_NO_LDS | E.FLAG_NO_ORDER, E.FLAG_NO_LDS | E.FLAG_NO_CULL):
        m = make_engine(rt, scene, W, H, S, B, cam, flags=flags)
        m.keep_samples(True)
        frame_m = m.render_frame(9)
        got_m = m.read_samples()
        for name in ("primary_id", "primary_t", "rng_draws", "radiance"):
            assert_bits(got_m[name], ref[name], f"{case} read from memory (flags {flags}): {name}")
        assert_bits(frame_m, ref["rgba"], f"{case} read from memory (flags {flags}): frame")
        m.close()


# RT_EXTRA_SEEDS="a-b" adds the seeds a..b-1 for a wider sweep than the suite's eight scenes
def _extra_seeds():
    spec = os.environ.get("RT_EXTRA_SEEDS", "")
    if "-" not in spec:
        return []
    lo, hi = spec.split("-", 1)
    return list(range(int(lo), int(hi)))


@pytest.mark.parametrize("seed", [111, 222, 333, 444, 555, 666, 777, 888] + _extra_seeds())
def test_random_scenes_without_fog_bit_exact(rt, orc, assets, seed, monkeypatch):
    """the same without fog volumes, so that the memory-resident variants below run the engine's own hierarchy over the
    meshes, the near-side-first walk, culling and pair steps (a scene with fog keeps the reference's order throughout)"""
    _random_scene_check(rt, orc, assets, seed, monkeypatch, fog=False)


@pytest.mark.parametrize("seed", [101, 202, 303, 404, 505, 606, 707, 808] + _extra_seeds())
def test_random_scenes_bit_exact(rt, orc, assets, seed, monkeypatch):
    """scenes nobody arranged: ids, distances, draw counts, radiance and frame bytes against the oracle, bit for bit"""
    _random_scene_check(rt, orc, assets, seed, monkeypatch, fog=True)


def _random_scene_check(rt, orc, assets, seed, monkeypatch, fog):
    tex, mat, obj, plane, (pos, look) = random_tables(seed, fog)
    assert fog or not (obj[:, 0] == 4).any()
    args = (tex, mat, obj, plane, assets["verts"], assets["tex"], assets["sky"], 0xC0FFEE + seed)
    scene = build_table_scene(rt.Scene(), *args)
    osc = build_table_scene(orc.Scene(orc.MATH_PORTABLE), *args)
    for a, b, n in zip(scene.dump_top(), osc.dump_top(), ("boxes", "left", "right", "leaf")):
        assert_bits(a, b, f"random scene {seed}: BVH {n}")
    W, H, S, B = 96, 54, 2, 5
    cam = rt.camera_look(pos, look)
    e = make_engine(rt, scene, W, H, S, B, cam)
    e.keep_samples(True)
    frame = e.render_frame(seed)
    got = e.read_samples()
    ref = osc.render(cam, W, H, S, B, seed, threads=8, want=("radiance", "rgba", "rng_draws", "primary_id", "primary_t"))
    assert_bits(got["primary_id"], ref["primary_id"], f"random scene {seed}: ids")
    assert_bits(got["primary_t"], ref["primary_t"], f"random scene {seed}: distances")
    assert_bits(got["rng_draws"], ref["rng_draws"], f"random scene {seed}: RNG draws")
    assert_bits(got["radiance"], ref["radiance"], f"random scene {seed}: radiance")
    assert_bits(frame, ref["rgba"], f"random scene {seed}: frame")
    # the same scene read from device memory instead of on-chip memory (the kernels of large scenes, here with mesh
    # instances, rotations and fog in them), with and without distance culling
    # ... and from on-chip memory with the 768-thread blocks that large batches use for their first bounce passes
    monkeypatch.setenv("RT_WIDE_BLOCKS", "1")
    # (scenes without fog read from memory are walked near side first with a stack; FLAG_NO_ORDER selects the
    # reference's order without one, as scenes with fog are always walked)
    E = rt.engine
    # (FLAG_NO_REGROUP: the reference's hierarchy above the lowest nodes of the meshes' trees instead of the engine's own)
    # (FLAG_NO_PAIRS: one box per step of the near-side-first walk instead of a node's two children per step)
    for flags in (E.FLAG_NO_LDS, E.FLAG_NO_LDS | E.FLAG_NO_CULL, E.FLAG_NO_LDS | E.FLAG_NO_ORDER, E.FLAG_NO_LDS | E.FLAG_NO_ORDER | E.FLAG_NO_CULL, 0,
                  E.FLAG_NO_REGROUP, E.FLAG_NO_REGROUP | E.FLAG_NO_LDS, E.FLAG_NO_LDS | E.FLAG_NO_PAIRS, E.FLAG_NO_LDS | E.FLAG_NO_PAIRS | E.FLAG_NO_CULL,
                  E.FLAG_NO_LDS | E.FLAG_NO_PAIRS | E.FLAG_NO_REGROUP, E.FLAG_NO_WIDE):    # (FLAG_NO_WIDE: on chip, two children per node throughout)
        m = make_engine(rt, scene, W, H, S, B, cam, flags=flags)
        m.keep_samples(True)
        frame_m = m.render_frame(seed)
        got_m = m.read_samples()
        for name in ("primary_id", "primary_t", "rng_draws", "radiance"):
            assert_bits(got_m[name], ref[name], f"random scene {seed} read from memory (flags {flags}): {name}")
        assert_bits(frame_m, ref["rgba"], f"random scene {seed} read from memory (flags {flags}): frame")
        m.close()


@pytest.mark.parametrize("seed", [1, 2])
def test_random_large_scenes_read_from_memory(rt, orc, assets, seed):
    """6 000 spheres (a fifth of them with negative radius: boxes with min > max all over the tree) and cubes and two
    meshes: a top-level tree deeper than the table of upper levels that the blocks keep on chip, so the walk crosses from
    table records to records in memory and, at every box with min > max, from records to entries and back.  Ids,
    distances, draw counts, radiance and frame against the oracle bit for bit, through the engine's shortcuts."""
    r = np.random.default_rng(9000 + seed)
    n = 6000
    tex = np.array([[0, *r.uniform(0.2, 1.0, 3), 0, 0, 0], [1, 0.9, 0.9, 0.9, 0.2, 0.5, 0.3], [2, 0, 0, 0, 0, 0, 0]], np.float32)
    mat = np.array([[1, 0, 0, 0, 0], [2, 1, 0, 0, 0], [1, 2, 0, 0, 0], [1, 1, 0, 0, 0]], np.float32)
    obj = np.zeros((n + 2, 10), np.float32)
    c = r.uniform([-60, 0.5, -40], [60, 30, 80], (n, 3))
    size = r.uniform(0.3, 2.0, n)
    cube = r.random(n) < 0.3
    neg = (~cube) & (r.random(n) < 0.2)
    obj[:n, 0] = np.where(cube, 2, 0)
    obj[:n, 1:4] = c
    obj[:n, 4] = np.where(neg, -size, size)
    obj[:n, 9] = r.choice([0, 1, 3], n)
    obj[n] = [3, 14.0, 1, 40.0, -8.0, 7.0, 5.0, 1, 0, 2]
    obj[n + 1] = [3, 9.0, 0, 90.0, 12.0, 5.0, 20.0, 1, 0, 0]
    args = (tex, mat, obj, [[0.0, 1]], assets["verts"], assets["tex"], assets["sky"], 555 + seed)
    scene = build_table_scene(rt.Scene(), *args)
    osc = build_table_scene(orc.Scene(orc.MATH_PORTABLE), *args)
    assert scene.flat_info()["entries"] > 10000
    W, H, S, B = 128, 72, 2, 4
    cam = rt.camera_look((0.0, 15.0, -70.0), (0.0, -0.05, 1.0))
    ref = osc.render(cam, W, H, S, B, seed, threads=8, want=("radiance", "rgba", "rng_draws", "primary_id", "primary_t"))
    assert (ref["primary_id"] >= 0).mean() > 0.5
    E = rt.engine
    for flags in (0, E.FLAG_NO_PAIRS, E.FLAG_NO_CULL, E.FLAG_NO_ORDER, E.FLAG_NO_REGROUP, E.FLAG_NO_PAIRS | E.FLAG_NO_CULL | E.FLAG_NO_REGROUP):
        e = make_engine(rt, scene, W, H, S, B, cam, flags=flags)
        e.keep_samples(True)
        frame = e.render_frame(seed)
        got = e.read_samples()
        for name in ("primary_id", "primary_t", "rng_draws", "radiance"):
            assert_bits(got[name], ref[name], f"large random scene {seed} (flags {flags}): {name}")
        assert_bits(frame, ref["rgba"], f"large random scene {seed} (flags {flags}): frame")
        e.close()


@pytest.mark.parametrize("n_tris", [1, 2, 3, 5, 8])
def test_tiny_meshes(rt, orc, assets, n_tris):
    """meshes of one to eight triangles (a tree that is one lowest node; a root whose children are lowest nodes; the
    smallest trees with nodes above those), twice in a scene with a few spheres, rotated and moved: on-chip and read from
    memory, through every shortcut, against the oracle bit for bit"""
    r = np.random.default_rng(1000 + n_tris)
    tris = []
    for k in range(n_tris):
        c = r.uniform([-1, -1, -0.3], [1, 1, 0.3])
        P = c + r.uniform(-0.8, 0.8, (3, 3))
        nrm = np.cross(P[1] - P[0], P[2] - P[0])
        nrm /= np.linalg.norm(nrm)
        for v in range(3):
            tris.append([*P[v], *r.uniform(0, 1, 2), *nrm])
    verts = np.array(tris, np.float32)
    tex = np.array([[0, 0.8, 0.7, 0.6, 0, 0, 0], [1, 0.9, 0.9, 0.9, 0.2, 0.5, 0.3], [2, 0, 0, 0, 0, 0, 0]], np.float32)
    mat = np.array([[1, 0, 0, 0, 0], [2, 1, 0, 0, 0], [1, 2, 0, 0, 0]], np.float32)
    obj = [[3, 6.0, 1, 20.0, -2.5, 3.0, 2.0, 1, 0, 0], [3, 5.0, 0, 75.0, 3.0, 2.5, 4.0, 1, 0, 2]]
    for _ in range(5):
        c = r.uniform([-6, 0.5, -1], [6, 6, 9])
        obj.append([0, *c, r.uniform(0.4, 1.5), 0, 0, 0, 0, int(r.integers(0, 2))])
    obj = np.array(obj, np.float32)
    args = (tex, mat, obj, [[0.0, 1]], verts, assets["tex"], assets["sky"], 4242 + n_tris)
    scene = build_table_scene(rt.Scene(), *args)
    osc = build_table_scene(orc.Scene(orc.MATH_PORTABLE), *args)
    W, H, S, B = 96, 54, 2, 4
    cam = rt.camera_look((0.0, 3.0, -9.0), (0.0, 0.0, 1.0))
    ref = osc.render(cam, W, H, S, B, 5, threads=8, want=("radiance", "rgba", "rng_draws", "primary_id", "primary_t"))
    assert ((ref["primary_id"] == 0) | (ref["primary_id"] == 1)).mean() > 0.002         # the meshes are in view
    E = rt.engine
    for flags in (0, E.FLAG_NO_WIDE, E.FLAG_NO_REGROUP, E.FLAG_NO_LDS, E.FLAG_NO_LDS | E.FLAG_NO_PAIRS, E.FLAG_NO_LDS | E.FLAG_NO_ORDER, E.FLAG_NO_LDS | E.FLAG_NO_CULL,
                  E.FLAG_NO_LDS | E.FLAG_NO_REGROUP, E.FLAG_NO_LDS | E.FLAG_NO_REGROUP | E.FLAG_NO_PAIRS):
        e = make_engine(rt, scene, W, H, S, B, cam, flags=flags)
        e.keep_samples(True)
        frame = e.render_frame(5)
        got = e.read_samples()
        for name in ("primary_id", "primary_t", "rng_draws", "radiance"):
            assert_bits(got[name], ref[name], f"{n_tris}-triangle meshes (flags {flags}): {name}")
        assert_bits(frame, ref["rgba"], f"{n_tris}-triangle meshes (flags {flags}): frame")
        e.close()


def test_large_mesh_read_from_memory(rt, orc, assets):
    """A mesh of 9 800 triangles (a bumpy sheet with the tree's cut-out texture, rotated and moved) among spheres: too large
    for on-chip memory, so the mesh's tree is walked by the kernels of large scenes, regrouped (top-down cuts, then
    bottom-up joins: more than 256 lowest nodes), behind its entry box, near side first.  Ids, distances, draw counts,
    radiance and frame against the oracle bit for bit, through every combination of the engine's own shortcuts."""
    r = np.random.default_rng(31)
    n = 70
    gx, gz = np.meshgrid(np.linspace(-1, 1, n + 1), np.linspace(-1, 1, n + 1), indexing="ij")
    gy = 0.15 * np.sin(5 * gx) * np.cos(4 * gz) + 0.03 * r.normal(size=gx.shape)
    P = np.stack([gx, gy, gz], axis=-1).astype(np.float32)
    UV = np.stack([(gx + 1) / 2, (gz + 1) / 2], axis=-1).astype(np.float32)
    tris = []
    for i in range(n):
        for j in range(n):
            for a, b, c in (((i, j), (i + 1, j), (i, j + 1)), ((i + 1, j), (i + 1, j + 1), (i, j + 1))):
                nrm = np.cross(P[b] - P[a], P[c] - P[a])
                nrm = (nrm / np.linalg.norm(nrm)).astype(np.float32)
                for v in (a, b, c):
                    tris.append([*P[v], *UV[v], *nrm])
    verts = np.array(tris, np.float32)
    assert len(verts) == 3 * 2 * n * n
    tex = np.array([[0, 0.8, 0.7, 0.6, 0, 0, 0], [1, 0.9, 0.9, 0.9, 0.2, 0.5, 0.3], [2, 0, 0, 0, 0, 0, 0]], np.float32)
    mat = np.array([[1, 0, 0, 0, 0], [2, 1, 0, 0, 0], [1, 2, 0, 0, 0]], np.float32)
    obj = [[3, 24.0, 1, 30.0, 1.0, 4.0, 6.0, 1, 0, 2]]                 # the mesh: size, rotation about y, offset, image texture
    for _ in range(24):
        c = r.uniform([-12, 0.5, -4], [12, 9, 18])
        obj.append([0, *c, r.uniform(0.4, 2.0), 0, 0, 0, 0, int(r.integers(0, 2))])
    obj = np.array(obj, np.float32)
    args = (tex, mat, obj, [[0.0, 1]], verts, assets["tex"], assets["sky"], 909)
    scene = build_table_scene(rt.Scene(), *args)
    osc = build_table_scene(orc.Scene(orc.MATH_PORTABLE), *args)
    info = scene.regroup_info()
    assert info["has_regrouped_layout"] == 1 and info["lowest_nodes"] > 4000 and info["mesh_tests_regrouped"] < info["mesh_tests_reference"]
    assert scene.flat_info()["entries"] * 32 > 80 * 1024               # not a scene for on-chip memory
    W, H, S, B = 96, 54, 2, 4
    cam = rt.camera_look((0.5, 13.0, -12.0), (0.0, -0.55, 1.0))
    ref = osc.render(cam, W, H, S, B, 21, threads=8, want=("radiance", "rgba", "rng_draws", "primary_id", "primary_t"))
    assert (ref["primary_id"] == 0).mean() > 0.08                       # the sheet is in view (much of it cut out by its texture)
    E = rt.engine
    counts = {}
    for flags in (0, E.FLAG_NO_REGROUP, E.FLAG_NO_ORDER, E.FLAG_NO_CULL, E.FLAG_NO_REGROUP | E.FLAG_NO_ORDER | E.FLAG_NO_CULL,
                  E.FLAG_NO_PAIRS, E.FLAG_NO_PAIRS | E.FLAG_NO_REGROUP, E.FLAG_NO_PAIRS | E.FLAG_NO_CULL):
        e = make_engine(rt, scene, W, H, S, B, cam, flags=flags)
        e.keep_samples(True)
        frame = e.render_frame(21)
        got = e.read_samples()
        for name in ("primary_id", "primary_t", "rng_draws", "radiance"):
            assert_bits(got[name], ref[name], f"large mesh (flags {flags}): {name}")
        assert_bits(frame, ref["rgba"], f"large mesh (flags {flags}): frame")
        counts[flags] = e.stats()["box_tests"]
        e.close()
    assert counts[0] < counts[E.FLAG_NO_REGROUP], counts
    print(f"large mesh, slab tests per frame by flags: {counts}")


def _culling_stress_rays(obj, rng, n_targets):
    """rays aimed at the silhouettes of spheres (where the computed distance is worst conditioned), from far away, a share
    of them nearly parallel to a coordinate axis (where a box face is crossed slowly), some with directions that are not
    of unit length (the reference's sphere formula assumes it), plus plain random rays"""
    sph = obj[obj[:, 0] == 0]
    pick = sph[rng.integers(0, len(sph), n_targets)]
    c, r = pick[:, 1:4].astype(np.float64), np.abs(pick[:, 4:5].astype(np.float64))
    dist = rng.uniform(5.0, 400.0, (n_targets, 1))
    u = rng.normal(size=(n_targets, 3))
    k = n_targets // 3
    axis = rng.integers(0, 3, k)
    u[:k] = rng.normal(size=(k, 3)) * np.exp(rng.uniform(np.log(1e-7), np.log(1e-2), (k, 1)))
    u[np.arange(k), axis] = rng.choice([-1.0, 1.0], k)
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    o = c - u * dist
    p = np.cross(u, rng.normal(size=(n_targets, 3)))
    p /= np.linalg.norm(p, axis=1, keepdims=True)
    eps = rng.choice([0.0, 1e-7, -1e-7, 1e-6, -1e-6, 1e-5, -1e-5, 1e-4, -1e-4, 1e-3, -1e-3, -0.5, -0.9], (n_targets, 1))
    d = c + r * (1.0 + eps) * p - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d32 = d.astype(np.float32)
    d32 /= np.sqrt((d32.astype(np.float64) ** 2).sum(axis=1, keepdims=True)).astype(np.float32)
    scale = np.ones((n_targets, 1), np.float32)
    scale[::7] = np.float32(1.0 + 3e-4)          # |d.d - 1| below the culling limit, far above rounding
    scale[3::11] = np.float32(0.9)               # above it: such a ray culls nothing
    d32 *= scale
    ro = rng.uniform([-150, 0.5, -200], [150, 60, 150], (n_targets // 2, 3)).astype(np.float32)
    rd = rng.normal(size=(n_targets // 2, 3)).astype(np.float32)
    rd /= np.linalg.norm(rd, axis=1, keepdims=True).astype(np.float32)
    return np.concatenate([o.astype(np.float32), ro]), np.concatenate([d32, rd])


def test_distance_culling_changes_nothing(rt, orc, assets):
    """Boxes behind the closest hit are skipped in scenes too large for on-chip memory.  The skipped boxes must be exactly
    those that cannot matter: ids, distances and radiance with culling, without it (RT_FLAG_NO_CULL) and from the oracle,
    which like the reference visits everything, are compared bit for bit on rays chosen to sit on the culling bound's
    weak spots, in a scene with spheres from 1 mm to 30 m next to each other."""
    rng = np.random.default_rng(77)
    obj, tex, mat = synthetic_tables(3000)
    extra = np.zeros((300, 10), np.float32)
    extra[:, 0] = 0
    extra[:, 1:4] = rng.uniform([-100, 1, -100], [100, 41, 100], (300, 3))
    extra[:200, 4] = np.exp(rng.uniform(np.log(1e-3), np.log(5e-2), 200))
    extra[200:250, 4] = rng.uniform(3.0, 30.0, 50)
    extra[250:, 0] = 2
    extra[250:, 4] = rng.uniform(0.05, 6.0, 50)
    extra[:, 9] = rng.integers(0, 3000, 300)
    obj = np.concatenate([obj, extra])
    plane = [[0.0, len(mat) - 1]]
    # a build key under which the first leaf of the tree (the one whose rejected HitInfo.distance survives) is a sphere
    for key in range(4242, 4300):
        args = (tex, mat, obj, plane, None, None, assets["sky"], key)
        scene = build_table_scene(rt.Scene(), *args)
        first = int(scene.dump_top()[3][0])
        if obj[first, 0] == 0 and obj[first, 4] > 0.3:
            break
    else:
        raise AssertionError("no build key puts a sphere first")
    osc = build_table_scene(orc.Scene(orc.MATH_PORTABLE), *args)
    assert scene.flat_info()["device_bytes"] > 160 * 1024
    cam = rt.camera_look((0.0, 20.0, -150.0), (0.0, 0.0, 1.0))
    e = make_engine(rt, scene, 256, 256, 2, 4, cam)
    e0 = make_engine(rt, scene, 256, 256, 2, 4, cam, flags=rt.engine.FLAG_NO_CULL)
    o, d = _culling_stress_rays(obj, rng, 60000)
    # rays that leave the first leaf's surface downwards: that sphere writes a negative HitInfo.distance and rejects; many
    # of these rays cross the plane and then meet the underground part of a large sphere — a hit the plane bound may
    # skip only if the stale value is honoured (the plane must still win, as in the reference)
    c0, r0 = obj[first, 1:4].astype(np.float64), float(obj[first, 4])
    nrm = rng.normal(size=(20000, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    so = c0 + nrm * r0 * 1.0001
    sd = nrm + rng.normal(size=(20000, 3)) * 0.7
    sd[:, 1] = -np.abs(sd[:, 1]) - 0.05
    sd /= np.linalg.norm(sd, axis=1, keepdims=True)
    sd32 = sd.astype(np.float32)
    sd32 /= np.sqrt((sd32.astype(np.float64) ** 2).sum(axis=1, keepdims=True)).astype(np.float32)
    keep = (sd32.astype(np.float64) * nrm).sum(axis=1) > 0.05            # pointing away from the sphere
    so, sd32 = so[keep].astype(np.float32), sd32[keep]
    osc_np = build_table_scene(orc.Scene(orc.MATH_PORTABLE), tex, mat, obj, np.zeros((0, 2), np.float32), None, None, assets["sky"], key)
    hit_np, rec_np = osc_np.trace(so, sd32)
    t_plane = -(so[:, 1] - np.float32(0.001)) / sd32[:, 1]
    n_trap = int(((hit_np >= 0) & (rec_np[:, 0] > t_plane * 1.01) & (t_plane > 0.001)).sum())
    assert n_trap > 50, f"only {n_trap} rays have their only hits beyond the plane"
    assert ((osc.trace(so, sd32)[0] == -1) & (osc.trace(so, sd32)[1][:, 0] < 0)).sum() > 50     # stale values do occur
    o, d = np.concatenate([o, so]), np.concatenate([d, sd32])
    hit, rec = e.trace(o, d)
    hit0, rec0 = e0.trace(o, d)
    hit_o, rec_o = osc.trace(o, d)
    assert (hit >= 0).mean() > 0.5
    assert_bits(hit, hit0, "ids with and without culling")
    assert_bits(rec, rec0, "hit records with and without culling")
    assert_bits(hit, hit_o, "ids vs the oracle")
    assert_bits(rec[:, 0], rec_o[:, 0], "distances vs the oracle")
    keys = (np.arange(len(o), dtype=np.uint64) * 2654435761 % (1 << 32)).astype(np.uint32)
    col = e.ray_color(o, d, 4, keys)
    assert_bits(col, e0.ray_color(o, d, 4, keys), "radiance with and without culling")
    assert_bits(col, osc.ray_color(o, d, 4, keys), "radiance vs the oracle")
    # the same rays through the kernels that take a node's children in the reference's order (no stack); e and e0 above
    # visit the near child first
    e1 = make_engine(rt, scene, 256, 256, 2, 4, cam, flags=rt.engine.FLAG_NO_ORDER)
    assert_bits(e1.ray_color(o, d, 4, keys), col, "radiance in the reference's child order")
    # and it does skip work: a frame's executed slab tests drop well below what the reference algorithm performs, and
    # further when the near child is visited first
    a = e.render_frame(5)
    n_cull = e.stats()["box_tests"]
    a0 = e0.render_frame(5)
    n_all = e0.stats()["box_tests"]
    a1 = e1.render_frame(5)
    n_unordered = e1.stats()["box_tests"]
    assert_bits(a, a0, "frame with and without culling")
    assert_bits(a, a1, "frame near side first and in the reference's order")
    # ... and through the reference's own hierarchy, every box the reference visits (no regrouped upper levels, no culling)
    e2 = make_engine(rt, scene, 256, 256, 2, 4, cam, flags=rt.engine.FLAG_NO_CULL | rt.engine.FLAG_NO_REGROUP)
    assert_bits(e2.ray_color(o, d, 4, keys), col, "radiance through the reference's hierarchy")
    a2 = e2.render_frame(5)
    n_ref = e2.stats()["box_tests"]
    assert_bits(a, a2, "frame through the regrouped and through the reference's hierarchy")
    assert 0 < n_cull < n_unordered < n_all <= n_ref and n_unordered < 0.7 * n_ref, (n_cull, n_unordered, n_all, n_ref)
    # ... and near side first one box per step (e above tests a node's two children per step)
    e3 = make_engine(rt, scene, 256, 256, 2, 4, cam, flags=rt.engine.FLAG_NO_PAIRS)
    hit3, rec3 = e3.trace(o, d)
    assert_bits(hit3, hit, "ids, one box per step")
    assert_bits(rec3, rec, "hit records, one box per step")
    assert_bits(e3.ray_color(o, d, 4, keys), col, "radiance, one box per step")
    assert_bits(e3.render_frame(5), a, "frame, one box per step")
    assert 0 < e3.stats()["box_tests"] < n_unordered
    print(f"slab tests per frame: {n_ref} in the reference's hierarchy, {n_all} regrouped, {n_unordered} culled in the stored child order, {n_cull} culled near side first")


@pytest.fixture(scope="module")
def c3_scene(rt, golden):
    """the 100 000-primitive traversal-stress scene at its real size, built through the C ABI"""
    import time
    from raytracer_amd.synthetic import synthetic_scene
    g = golden("scene_c3")
    t0 = time.perf_counter()
    scene = synthetic_scene(rt, int(g["n"]), int(g["build_key"]))
    return scene, time.perf_counter() - t0


def _sha(a):
    import hashlib
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def test_c3_tree_is_the_reference_tree(rt, golden, c3_scene):
    """BVH_Node::BVH_Node (Object.cpp:181-225) over 100 000 objects: boxes, links and leaf order of the tree the reference
    built from the same list under the same key (digests of the whole arrays, every 1000th row for a readable failure)"""
    g = golden("scene_c3")
    scene, seconds = c3_scene
    boxes, left, right, leaf = scene.dump_top()
    assert len(left) == int(g["n_inner"])
    for name, got in (("boxes", boxes), ("left", left), ("right", right), ("leaf_ids", leaf)):
        assert_bits(got[::1000], g[f"top_{name}_rows"], f"C3 tree {name}, every 1000th row")
        assert_bits(_sha(got), g[f"top_{name}_sha256"], f"C3 tree {name}, digest")
    info = scene.flat_info()
    assert info["entries"] > 2 * int(g["n"])            # too large for LDS: the traversal reads it from memory
    print(f"C3 scene: {info['entries']} entries, built in {seconds:.2f} s (reference: {float(g['reference_build_seconds']):.1f} s)")


def test_c3_map_and_crop_vs_reference(rt, orc, golden, assets, c3_scene):
    """sample (0,0) of a 480 x 270 frame and a 32 x 32 crop of the 3840 x 2160, 256 spp frame: ids, distances and draw
    counts equal the reference's bit for bit; radiance and colour equal the portable oracle's bit for bit"""
    g = golden("scene_c3")
    scene, _ = c3_scene
    cam = rt.camera_look(g["cam_pos"], g["cam_look"])
    seed = int(g["seed"])
    Wm, Hm, Sm, Bm = (int(v) for v in g["map_cfg"])
    e = make_engine(rt, scene, Wm, Hm, Sm, Bm, cam)
    e.keep_samples(True)
    e.render(seed)
    got = e.read_samples(want=("primary_id", "primary_t"))
    assert_bits(got["primary_id"].reshape(Hm, Wm), g["map_primary_id"], "C3 map ids vs the reference")
    assert_bits(got["primary_t"].reshape(Hm, Wm), g["map_primary_t"], "C3 map distances vs the reference")
    assert len(np.unique(g["map_primary_id"])) > 5000     # the map sees thousands of different objects
    e.close()
    W, H, S, B = (int(v) for v in g["cfg"])
    rect = tuple(int(v) for v in g["crop"])
    e = make_engine(rt, scene, W, H, S, B, cam, max_batch_paths=1 << 22)
    e.keep_samples(True)
    e.set_window(rect)
    e.render(seed)
    frame = e.read_frame()
    got = e.read_samples()
    x0, y0, x1, y1 = rect
    assert_bits(got["primary_id"], g["crop_primary_id"], "C3 crop ids vs the reference")
    assert_bits(got["primary_t"], g["crop_primary_t"], "C3 crop distances vs the reference")
    assert_bits(got["rng_draws"].astype(np.uint16), g["crop_rng_draws"], "C3 crop RNG draws vs the reference")
    assert_frame_close(frame[y0:y1, x0:x1], g["crop_rgba"], "C3 crop colour vs the reference")
    sub = g["crop_radiance_pixels"]
    same = np.all(got["radiance"][sub].view(np.uint32) == g["crop_radiance"].view(np.uint32), axis=2).mean()
    assert same > 0.995, f"C3 crop radiance identical to the reference's for {same:.4f} of samples"
    obj, tex, mat = synthetic_tables(int(g["n"]))
    osc = build_table_scene(orc.Scene(orc.MATH_PORTABLE), tex, mat, obj, [[0.0, len(mat) - 1]], None, None, assets["sky"], int(g["build_key"]))
    ref = osc.render(cam, W, H, S, B, seed, rect=rect, threads=16, want=("radiance", "rgba", "rng_draws"))
    assert_bits(got["radiance"], ref["radiance"], "C3 crop radiance vs the oracle")
    assert_bits(got["rng_draws"], ref["rng_draws"], "C3 crop RNG draws vs the oracle")
    assert_bits(frame[y0:y1, x0:x1], ref["rgba"], "C3 crop colour vs the oracle")


@pytest.mark.parametrize("n_ranks", [2, 4, 8])
def test_c4_rank_split_of_the_large_scene(rt, golden, c3_scene, n_ranks):
    """C4's workload (the 100 000-primitive scene, tiles interleaved over 2 / 4 / 8 ranks) at a reduced frame size: the
    shares assembled are the one-rank frame byte for byte, and the shares' ray counts add up to the frame's"""
    import torch
    g = golden("scene_c3")
    scene, _ = c3_scene
    cam = rt.camera_look(g["cam_pos"], g["cam_look"])
    W, H, S, B = 640, 360, 2, 8
    one = make_engine(rt, scene, W, H, S, B, cam)
    base = one.render_frame(int(g["seed"]))
    rays = one.stats()["rays"]
    e = make_engine(rt, scene, W, H, S, B, cam)
    nb = e.tile_bytes(n_ranks)
    gathered = torch.zeros(n_ranks * nb, dtype=torch.uint8, device="cuda")
    share = []
    for r in range(n_ranks):
        e.render(int(g["seed"]), rank=r, n_ranks=n_ranks, d_out=gathered.data_ptr() + r * nb)
        share.append(e.stats()["rays"])
    out = torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda")
    e.assemble(gathered.data_ptr(), n_ranks, d_out=out.data_ptr())
    e.sync()
    assert_bits(out.cpu().numpy().reshape(H, W, 4), base, f"{n_ranks}-rank split of the large scene")
    assert sum(share) == rays
    assert max(share) < 1.5 * rays / n_ranks, f"shares {share} of {rays} rays: interleaved tiles should balance"


def test_c3_full_frame_properties(rt, golden, c3_scene):
    """BASELINE's full size (100 000 primitives, 3840 x 2160, 256 spp, 8 bounces; 16 batches of 2^27 paths): the frame is
    deterministic, a window renders the bytes of the frame, and the golden crop sits in it"""
    g = golden("scene_c3")
    scene, _ = c3_scene
    cam = rt.camera_look(g["cam_pos"], g["cam_look"])
    W, H, S, B = (int(v) for v in g["cfg"])
    seed = int(g["seed"])
    e = make_engine(rt, scene, W, H, S, B, cam)
    a = e.render_frame(seed)
    st = e.stats()
    assert st["paths"] == W * H * S * S and st["rays"] > st["paths"]
    assert e.cfg.max_batch_paths == 1 << 27 and -(-W * H * S * S // (1 << 27)) == 16      # 16 batches
    assert (a[..., 0] == 255).all()
    x0, y0, x1, y1 = (int(v) for v in g["crop"])
    assert_frame_close(a[y0:y1, x0:x1], g["crop_rgba"], "C3 full frame, crop region vs the reference")
    # C4 at its real size: the frame as the eight shares of an 8-GPU node (tiles interleaved, two batches of 2^27 paths
    # per share), every share rendered here on the one GPU into its place in the gathered buffer, then scattered
    import torch
    n_ranks = 8
    nb = e.tile_bytes(n_ranks)
    gathered = torch.zeros(n_ranks * nb, dtype=torch.uint8, device="cuda")
    share = []
    for r in range(n_ranks):
        e.render(seed, rank=r, n_ranks=n_ranks, d_out=gathered.data_ptr() + r * nb)
        share.append(e.stats()["rays"])
    out = torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda")
    e.assemble(gathered.data_ptr(), n_ranks, d_out=out.data_ptr())
    e.sync()
    assert_bits(out.cpu().numpy().reshape(H, W, 4), a, "C4: eight shares of the full-size frame, assembled")
    assert sum(share) == st["rays"], "C4: the shares' rays add up to the frame's"
    assert max(share) < 1.1 * st["rays"] / n_ranks, f"C4 shares {share}: interleaved tiles should balance within 10 %"
    del gathered, out
    e.set_window((1800, 1000, 2056, 1128))
    c = e.render_frame(seed)
    assert_bits(c[1000:1128, 1800:2056], a[1000:1128, 1800:2056], "C3 window vs full frame")
    e.set_window(None)
    b = e.render_frame(seed)
    assert_bits(a, b, "two renders of the C3 frame")
    n_cull, rays = e.stats()["box_tests"], e.stats()["rays"]
    e.close()
    # the same frame visiting every box the reference visits
    e0 = make_engine(rt, scene, W, H, S, B, cam, flags=rt.engine.FLAG_NO_CULL)
    a0 = e0.render_frame(seed)
    n_all = e0.stats()["box_tests"]
    assert_bits(a0, a, "C3 frame without distance culling")
    print(f"C3 slab tests per ray: {n_cull / rays:.1f} executed with culling, {n_all / rays:.1f} without")
    assert n_cull < n_all
    e0.close()
    # ... and one box per step instead of a node's two children (a quarter of the frame's height is enough here)
    e1 = make_engine(rt, scene, W, H, S, B, cam, flags=rt.engine.FLAG_NO_PAIRS)
    e1.set_window((0, 810, W, 1350))
    a1 = e1.render_frame(seed)
    assert_bits(a1[810:1350], a[810:1350], "C3 rows 810-1349, one box per step")
    e1.close()
