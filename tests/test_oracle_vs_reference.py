"""The CPU oracle against the reference's own code on scenes nobody arranged.  oracle/_ref/libref.so is the reference's
C++ compiled in place (oracle/Makefile ref); where it has not been built these tests are skipped and the oracle stands
on the committed golden vectors alone (tests/test_oracle_golden.py)."""
import numpy as np
import pytest

from conftest import assert_bits, build_table_scene, random_tables
from oracle import refbind

pytestmark = pytest.mark.skipif(not refbind.available(), reason="oracle/_ref/libref.so has not been built")


@pytest.fixture(scope="module")
def ref():
    return refbind.Ref()


@pytest.mark.parametrize("seed", [101, 202, 303, 404, 505, 606, 707, 808, 909, 1010, 1111, 1212])
def test_random_scenes_oracle_is_the_reference(orc, ref, assets, seed):
    """same tables into World (through the harness) and into the oracle with libm math: BVH topology, primary ids and
    distances, RNG draw counts, per-sample radiance and frame bytes must agree bit for bit"""
    tex, mat, obj, plane, (pos, look) = random_tables(seed)
    key = 0xC0FFEE + seed
    w = ref.world(key)
    try:
        ref.set_custom(w, key, tex, mat, obj, plane if plane else np.zeros((0, 2), np.float32))
        osc = build_table_scene(orc.Scene(orc.MATH_LIBM), tex, mat, obj, plane, assets["verts"], assets["tex"], assets["sky"], key)
        for a, b, n in zip(ref.dump_top(w), osc.dump_top(), ("boxes", "left", "right", "leaf")):
            assert_bits(a, b, f"random scene {seed}: BVH {n}")
        W, H, S, B = 64, 36, 2, 5
        ref.set_camera(w, pos, look)
        want = ("radiance", "primary_id", "primary_t", "rng_draws", "rgba")
        a = ref.render(w, W, H, S, B, seed, threads=8, want=want)
        b = osc.render(orc.camera(pos, look), W, H, S, B, seed, threads=8, want=want)
        for k in want:
            assert_bits(a[k], b[k], f"random scene {seed}: {k}")
    finally:
        ref.free(w)


REF_ASSETS = "/root/reference/Raytracer/src/stb_image"


@pytest.mark.skipif(not __import__("os").path.isdir(REF_ASSETS), reason="the reference's asset directory is not here")
def test_asset_decoders_on_every_reference_asset(rt, ref):
    """the product's PNG decoder and OBJ parser against stb_image (DataUtility.cpp:17-28, 98-109) and LoadOBJ
    (Object.cpp:349-435) on every asset the reference ships, the ones its compiled-in scene does not use included
    (Earth.png, Rock.obj); three and four channels, rows flipped as the reference loads them"""
    import os
    E = rt.engine
    for name in ("Earth.png", "skybox12.png", "tree_texture.png"):
        for channels in (3, 4):
            want, want_fc = ref.load_image(os.path.join(REF_ASSETS, name), channels)
            got, got_fc = E.load_png(os.path.join(REF_ASSETS, name), channels)
            assert got_fc == want_fc, f"{name}: channels in the file"
            assert_bits(got, want, f"{name} decoded to {channels} channels")
    for name in ("Rock.obj", "tree.obj"):
        assert_bits(E.load_obj(os.path.join(REF_ASSETS, name)), ref.load_obj(os.path.join(REF_ASSETS, name)), f"{name} vertices")
