import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
ASSETS = os.path.join(ROOT, "assets", "src", "stb_image")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def bits_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and a.tobytes() == b.tobytes()


def assert_bits(a, b, what=""):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, f"{what}: {a.shape}/{a.dtype} vs {b.shape}/{b.dtype}"
    if a.tobytes() != b.tobytes():
        av = a.view(np.uint8).reshape(-1) if a.dtype.itemsize == 1 else a.reshape(-1).view(f"u{a.dtype.itemsize}")
        bv = b.view(np.uint8).reshape(-1) if b.dtype.itemsize == 1 else b.reshape(-1).view(f"u{b.dtype.itemsize}")
        bad = np.nonzero(av != bv)[0]
        raise AssertionError(f"{what}: {len(bad)} of {av.size} elements differ, first at {bad[:5]}: "
                             f"{a.reshape(-1)[bad[:5]]} vs {b.reshape(-1)[bad[:5]]}")


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + ".npz"))
        return cache[name]
    return load


@pytest.fixture(scope="session")
def orc():
    """the oracle binding; builds oracle/liboracle.so on first use when a compiler is present"""
    from oracle import orcbind
    if not os.path.exists(orcbind.LIB_PATH):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    orcbind.lib()
    return orcbind


@pytest.fixture(scope="session")
def rt():
    import raytracer_amd
    raytracer_amd.engine.lib()
    return raytracer_amd


@pytest.fixture(scope="session")
def assets(rt):
    """decoded default-scene assets (by the product's own decoders, which test_host_scene pins to the reference's)"""
    E = rt.engine
    verts = E.load_obj(os.path.join(ASSETS, "tree.obj"))
    tex, _ = E.load_png(os.path.join(ASSETS, "tree_texture.png"), 4)
    sky, _ = E.load_png(os.path.join(ASSETS, "skybox12.png"), 3)
    return {"verts": verts, "tex": tex, "sky": sky}


@pytest.fixture(scope="session")
def build_key(golden):
    return int(golden("scene_default")["build_key"])


def make_oracle_default(orc, assets, key, math_mode):
    s = orc.Scene(math_mode)
    s.default_scene(assets["verts"], assets["tex"], assets["sky"])
    s.build(key)
    return s


@pytest.fixture(scope="session")
def oracle_default_libm(orc, assets, build_key):
    return make_oracle_default(orc, assets, build_key, orc.MATH_LIBM)


@pytest.fixture(scope="session")
def oracle_default(orc, assets, build_key):
    """default scene in the oracle with the portable sinf/atan2f the HIP kernels use"""
    return make_oracle_default(orc, assets, build_key, orc.MATH_PORTABLE)


def build_table_scene(S, tex_t, mat_t, obj_t, plane_t, verts, tex_img, sky, key):
    """Build a scene from fixture tables (oracle/gen_golden.py custom-scene format) through either binding: S is a fresh
    oracle Scene or product Scene — both expose the same construction methods."""
    from raytracer_amd.synthetic import build_from_tables
    return build_from_tables(S, tex_t, mat_t, obj_t, plane_t, sky, key, verts=verts, tex_img=tex_img)


def synthetic_tables(n):
    from raytracer_amd.synthetic import synthetic_tables as st
    return st(n)
