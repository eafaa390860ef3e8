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


def random_tables(seed, fog=True):
    """a small scene drawn from a seeded generator: every primitive kind, texture kind, material, rotation axis and
    wrapper, at sizes and places that make them overlap; fog=False: no fog volumes (scenes with fog are walked in the
    reference's order by kernels of their own; without it the engine's shortcuts — its own hierarchy over meshes, near
    side first, culling, pair steps — are what runs), more objects and more meshes instead"""
    r = np.random.default_rng(seed)
    tex = [[0, *r.uniform(0.1, 1.0, 3), 0, 0, 0], [1, *r.uniform(0.1, 1.0, 3), *r.uniform(0.1, 1.0, 3)], [2, 0, 0, 0, 0, 0, 0]]
    mat = [[1, 0, 0, 0, 0], [1, 1, 0, 0, 0], [1, 2, 0, 0, 0], [2, 0, 0, 0, 0], [2, 1, 0, 0, 0], [3, 0, 0, 0, 0], [4, -1, *r.uniform(1.0, 6.0, 3)],
           [0, -1, 0, 0, 0]]
    obj = []
    for _ in range(int(r.integers(6, 40)) if fog else int(r.integers(2, 90))):
        k = int(r.choice([0, 0, 0, 1, 2, 4] if fog else [0, 0, 0, 1, 1, 2]))
        c = r.uniform(-12, 12, 3) * [1, 0.4, 1] + [0, 3, 8]
        m = int(r.integers(0, len(mat)))
        if k == 0:
            obj.append([0, *c, float(r.choice([1, 1, 1, -1])) * r.uniform(0.3, 3.0), 0, 0, 0, 0, m])
        elif k == 1:
            h = r.uniform(0.3, 2.5, 3)
            obj.append([1, *(c - h), *(c + h), 0, 0, m])
        elif k == 2:
            obj.append([2, *c, r.uniform(0.3, 2.5), 0, 0, 0, 0, m])
        else:
            obj.append([4, *c, r.uniform(1.0, 5.0), r.uniform(0.02, 0.6), int(r.integers(-1, 3)), r.uniform(0, 360),
                        float(r.choice([0.0, 0.0, r.uniform(-3, 3)])), 5])
    for _ in range(int(r.integers(0, 3)) if fog else int(r.integers(0, 5))):   # mesh instances: scale, rotation axis, degrees, offset, offset present, material (image texture)
        mv = r.uniform(-6, 6, 3) * [1, 0.2, 1] + [0, 2, 6]
        obj.append([3, r.uniform(2, 9), int(r.integers(-1, 3)), r.uniform(0, 360), *mv, int(r.integers(0, 2)), 0, int(r.choice([2, 2, 0, 3]))])
    plane = [[float(r.uniform(-1, 1)), int(r.choice([0, 1, 3, 4]))] for _ in range(int(r.integers(0, 3)))]
    cam = (r.uniform(-6, 6), r.uniform(1, 6), r.uniform(-14, -6)), (r.uniform(-0.3, 0.3), r.uniform(-0.2, 0.1), 1.0)
    return np.array(tex, np.float32), np.array(mat, np.float32), np.array(obj, np.float32), plane, cam
