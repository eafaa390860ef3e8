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


@pytest.mark.parametrize("seed,fog", [(s, True) for s in (101, 202, 303, 404, 505, 606, 707, 808, 909, 1010, 1111, 1212)] +
                         [(s, False) for s in (111, 222, 333, 444, 555, 666, 777, 888)])
def test_random_scenes_oracle_is_the_reference(orc, ref, assets, seed, fog):
    """same tables into World (through the harness) and into the oracle with libm math: BVH topology, primary ids and
    distances, RNG draw counts, per-sample radiance and frame bytes must agree bit for bit (with fog volumes, and the
    scenes without them — up to 90 objects and four meshes — that the GPU suite renders through the engine's shortcuts)"""
    tex, mat, obj, plane, (pos, look) = random_tables(seed, fog)
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


ODD_OBJ = (
    "# records the reference's parser reads in its own way\r\n"
    "v 0 0 0\r\n"
    "v 1.5e0 0 0 2\n"
    "\tv   0 +1 .5\n"
    "v -1 1e 7 8\n"                 # "1e" is not a number: y = 0 and the record ends (z keeps 0)
    "v 2 abc 3\n"                   # y = 0, z stays 0
    "v 3\n"                         # y, z missing
    "v 0.1 0.2 0.30000001192092896 trailing\n"
    "v 1e-3 -2.5E+1 1.e1\n"
    "v 4 - 5\n"
    "vt 0.25 0.75\n"
    "vt 1 0 0\n"
    "vt .5\n"
    "vn 0 0 2\n"
    "vn 1 1 0\n"
    "vn 0 3\n"
    "f 1/1/1 2/2/2 3/3/3\n"
    "f 1 2 3\n"                     # no texture coordinate, no normal: the face normal
    "f 1//2 2//2 3//2 4//2 5//2\n"  # a fan of three triangles
    "f -1/-1/-1 -2/-2/-2 -3/-3/-3\n"
    "f 6/1x/2 7/+2/3 8/3/1junk\n"
    "f 1/1/1 2/2/2\n"               # two corners: nothing
    "vn 0 0 0\n"                    # normalises to NaN
    "f 9/1/4 1/1/1 2/2/2\n"
    "g ignored 1 2 3\n"
    "f 3/3/3 2/2/2 1/1/1")          # no newline at the end


def test_obj_parser_on_odd_records(rt, ref, tmp_path):
    """fields that are missing, malformed, signed, relative, or separated by tabs and carriage returns: the product's
    tokenizer must hand out what LoadOBJ (Object.cpp:349-435: stream extraction, getline on '/', atoi) hands out"""
    path = tmp_path / "odd.obj"
    path.write_bytes(ODD_OBJ.encode())
    got, want = rt.engine.load_obj(str(path)), ref.load_obj(str(path))
    assert len(want) == 3 * 9
    assert_bits(got, want, "odd OBJ records")
