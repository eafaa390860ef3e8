"""Synthetic traversal-stress scenes: N spheres and axis-aligned boxes drawn from one PCG32 stream, the default metal
checker plane and the default skybox.  The object list goes through the same BVH construction as any other scene
(BVH_Node::BVH_Node, Object.cpp:181-225), so the reference builds the identical tree from the same list."""
import os

import numpy as np


class PCG32:
    """PCG XSH-RR 64/32 (O'Neill, pcg-random.org minimal C implementation)"""

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
    """Object i: even -> sphere(centre, size), odd -> cube(centre, half = size); centre uniform in
    [-100,100] x [1,41] x [-100,100], size uniform in [0.2,1.0]; material by i mod 4: 0,1 diffuse colour, 2 metal colour,
    3 diffuse checker(colour, white).  Returns (obj rows, texture rows, material rows); the last material is the metal
    checker of the default plane.  Row formats: obj = kind(0 sphere | 2 cube centre,half), 8 params, material;
    texture = kind, colour1, colour2; material = type, texture, emit."""
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


def build_from_tables(S, tex_t, mat_t, obj_t, plane_t, sky, key, verts=None, tex_img=None):
    """Fill and build scene S (anything with the construction methods of raytracer_amd.Scene) from table rows."""
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
        elif k == 4:    # fog: centre, radius, density, rotation axis (-1 none), degrees, x offset of a movement wrapper (0 none)
            S.fog(row[1:4], float(row[4]), float(row[5]), m, int(row[6]), float(row[7]), (float(row[8]), 0.0, 0.0) if row[8] != 0 else None)
        else:
            S.mesh(verts, float(row[1]), m, int(row[2]), float(row[3]), row[4:7] if row[7] != 0 else None)
    for p in plane_t:
        S.plane(float(p[0]), mt[int(p[1])])
    S.build(key)
    return S


def synthetic_scene(rt, n, key):
    E = rt.engine
    sky, _ = E.load_png(os.path.join(E.ASSET_DIR, "src", "stb_image", "skybox12.png"), 3)
    obj, tex, mat = synthetic_tables(n)
    return build_from_tables(rt.Scene(), tex, mat, obj, [[0.0, len(mat) - 1]], sky, key)
