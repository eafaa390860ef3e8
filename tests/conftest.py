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
    """Build a scene from the fixture tables (oracle/gen_golden.py custom-scene format) through either binding:
    S is a freshly created oracle Scene or product Scene — both expose the same construction methods."""
    S.skybox(sky)
    tx = []
    for t in tex_t:
        k = int(t[0])
        tx.append(S.texture_color(t[1:4]) if k == 0 else (S.texture_checker(t[1:4], t[4:7]) if k == 1 else S.texture_image(tex_img)))
    mt = [S.material(int(m[0]), tx[int(m[1])] if m[1] >= 0 else -1, m[2:5]) for m in mat_t]
    for row in obj_t:
        k = int(row[0])
        m = mt[int(row[9])]
        if k == 0:
            S.sphere(row[1:4], float(row[4]), m)
        elif k == 1:
            S.cube_minmax(row[1:4], row[4:7], m)
        elif k == 2:
            S.cube_center(row[1:4], float(row[4]), m)
        else:
            S.mesh(verts, float(row[1]), m, int(row[2]), float(row[3]), row[4:7] if row[7] != 0 else None)
    for p in plane_t:
        S.plane(float(p[0]), mt[int(p[1])])
    S.build(key)
    return S


# PCG32 XSH-RR 64/32 (pcg-random.org minimal C implementation): the stream synthetic scenes are drawn from
class PCG32:
    def __init__(self, seed=0x853C49E6748FEA9B, inc=0xDA3E39CB94B95BDB):
        self.state, self.inc = 0, (inc << 1 | 1) & (2**64 - 1)
        self.next()
        self.state = (self.state + seed) & (2**64 - 1)
        self.next()

    def next(self):
        old = self.state
        self.state = (old * 6364136223846793005 + self.inc) & (2**64 - 1)
        xs = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xs >> rot) | (xs << ((-rot) & 31))) & 0xFFFFFFFF

    def u01(self):
        return np.float32((self.next() >> 8) / 16777216.0)


def synthetic_tables(n, seed=0x853C49E6748FEA9B):
    """i even -> sphere, odd -> cube(centre, half); centre uniform in [-100,100]x[1,41]x[-100,100], size in [0.2,1.0];
    material by i mod 4: 0,1 diffuse colour, 2 metal colour, 3 diffuse checker; last material = the default metal plane."""
    g = PCG32(seed)
    obj = np.zeros((n, 10), np.float32)
    tex, mat = [], []
    f = np.float32
    for i in range(n):
        cx = f(-100) + f(200) * g.u01()
        cy = f(1) + f(40) * g.u01()
        cz = f(-100) + f(200) * g.u01()
        size = f(0.2) + f(0.8) * g.u01()
        col = [g.u01(), g.u01(), g.u01()]
        obj[i, :5] = [0 if i % 2 == 0 else 2, cx, cy, cz, size]
        m = i % 4
        tex.append([1, *col, 1, 1, 1] if m == 3 else [0, *col, 0, 0, 0])
        mat.append([2 if m == 2 else 1, len(tex) - 1, 0, 0, 0])
        obj[i, 9] = len(mat) - 1
    tex.append([1, 1, 1, 1, 0.2, 0.6, 0.3])
    mat.append([2, len(tex) - 1, 0, 0, 0])
    return obj, np.array(tex, np.float32), np.array(mat, np.float32)
