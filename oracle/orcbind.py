"""TEST INFRASTRUCTURE: ctypes binding of oracle/liboracle.so (rt_oracle.c, the plain-C restatement of the
reference's per-pixel path).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")

TEX_COLOR, TEX_CHECKER, TEX_IMAGE = 0, 1, 2
MAT_NONE, MAT_DIFFUSE, MAT_METAL, MAT_ISOTROPIC, MAT_LIGHT = 0, 1, 2, 3, 4
MATH_LIBM, MATH_PORTABLE = 0, 1
COUNTER_NAMES = ("rays", "aabb", "sph_test", "sph_hit", "cube_test", "cube_hit", "plane_test", "plane_hit",
                 "tri_test", "tri_uv", "inst_enter", "inst_hit", "samples", "shade", "rng", "fog_test")


def _opt(arr):
    return None if arr is None else arr.ctypes.data_as(C.c_void_p)


def _f3(v):
    return np.ascontiguousarray(v, np.float32).reshape(3)


_L = None


def lib():
    global _L
    if _L is not None:
        return _L
    L = C.CDLL(LIB_PATH)
    L.orc_scene_new.restype = C.c_void_p
    L.orc_scene_free.argtypes = [C.c_void_p]
    L.orc_set_math.argtypes = [C.c_void_p, C.c_int]
    L.orc_add_texture.restype = C.c_int
    L.orc_add_texture.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.orc_add_material.restype = C.c_int
    L.orc_add_material.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.orc_add_sphere.restype = C.c_int
    L.orc_add_sphere.argtypes = [C.c_void_p, f32p, C.c_float, C.c_int]
    L.orc_add_cube_minmax.restype = C.c_int
    L.orc_add_cube_minmax.argtypes = [C.c_void_p, f32p, f32p, C.c_int]
    L.orc_add_cube_center.restype = C.c_int
    L.orc_add_cube_center.argtypes = [C.c_void_p, f32p, C.c_float, C.c_int]
    L.orc_add_fog.restype = C.c_int
    L.orc_add_fog.argtypes = [C.c_void_p, f32p, C.c_float, C.c_float, C.c_int, C.c_int, C.c_float, C.c_int, f32p]
    L.orc_pm_logf.restype = C.c_float
    L.orc_pm_logf.argtypes = [C.c_float]
    L.orc_add_mesh.restype = C.c_int
    L.orc_add_mesh.argtypes = [C.c_void_p, f32p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_float, C.c_int, f32p]
    L.orc_add_plane.restype = C.c_int
    L.orc_add_plane.argtypes = [C.c_void_p, C.c_float, C.c_int]
    L.orc_set_skybox.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int]
    L.orc_default_scene.argtypes = [C.c_void_p, f32p, C.c_int, u8p, C.c_int, C.c_int, u8p, C.c_int, C.c_int]
    L.orc_build.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_camera.argtypes = [f32p, f32p, f32p]
    L.orc_camera_input.argtypes = [f32p, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int]
    L.orc_dump_top.restype = C.c_int
    L.orc_dump_top.argtypes = [C.c_void_p, C.c_int, f32p, i32p, i32p, i32p, i32p]
    L.orc_dump_mesh.restype = C.c_int
    L.orc_dump_mesh.argtypes = [C.c_void_p, C.c_int, C.c_int, f32p, i32p, i32p, C.c_int, f32p, i32p]
    L.orc_dump_objects.restype = C.c_int
    L.orc_dump_objects.argtypes = [C.c_void_p, C.c_int, f32p]
    L.orc_dump_instance.restype = C.c_int
    L.orc_dump_instance.argtypes = [C.c_void_p, C.c_int, f32p]
    L.orc_aabb_hit.argtypes = [f32p, C.c_int, f32p, f32p, u8p]
    L.orc_normalize.argtypes = [C.c_int, f32p, f32p]
    L.orc_intersect_prim.argtypes = [C.c_int, f32p, C.c_int, f32p, f32p, C.c_float, i32p, f32p]
    L.orc_intersect_tri.argtypes = [f32p, C.c_void_p, C.c_int, C.c_int, C.c_int, f32p, f32p, C.c_float, i32p, f32p]
    L.orc_trace.argtypes = [C.c_void_p, C.c_int, f32p, f32p, i32p, f32p]
    L.orc_ray_color.argtypes = [C.c_void_p, C.c_int, f32p, f32p, C.c_int, u32p, C.c_float, f32p]
    L.orc_skybox.argtypes = [C.c_void_p, C.c_int, f32p, f32p]
    L.orc_render.argtypes = [C.c_void_p, i32p, f32p, f32p, C.c_uint32] + [C.c_int] * 5 + [C.c_void_p] * 6
    L.orc_render_pixel_keyed.argtypes = [C.c_void_p, i32p, f32p, f32p, C.c_uint32] + [C.c_int] * 4 + [C.c_void_p]
    L.orc_render_framemanager.argtypes = [C.c_void_p, i32p, f32p, f32p, C.c_int, C.c_int, C.c_void_p]
    L.orc_thread_pixel_order.restype = C.c_int
    L.orc_thread_pixel_order.argtypes = [C.c_int] * 6 + [u32p]
    L.orc_counters.argtypes = [C.c_void_p, u64p, C.c_int]
    L.orc_repeat_counters.argtypes = [C.c_void_p, u64p]
    L.orc_rand31.restype = C.c_uint32
    L.orc_rand31.argtypes = [C.c_uint32, C.c_uint32]
    L.orc_sample_key.restype = C.c_uint32
    L.orc_sample_key.argtypes = [C.c_uint32] * 3
    L.orc_pm_sinf.restype = C.c_float
    L.orc_pm_sinf.argtypes = [C.c_float]
    L.orc_pm_atan2f.restype = C.c_float
    L.orc_pm_atan2f.argtypes = [C.c_float, C.c_float]
    L.orc_std_sort.argtypes = [C.c_int, f32p, i32p]
    _L = L
    return L


class Scene:
    """An oracle scene.  Objects added in creation order get stable ids 0,1,2,...; planes continue the numbering."""

    def __init__(self, math_mode=MATH_PORTABLE):
        self.L = lib()
        self.h = self.L.orc_scene_new()
        self.L.orc_set_math(self.h, math_mode)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_scene_free(self.h)
            self.h = None

    def set_math(self, mode):
        self.L.orc_set_math(self.h, mode)

    # -- construction ----------------------------------------------------------------------------
    def texture_color(self, c):
        return self.L.orc_add_texture(self.h, TEX_COLOR, _opt(_f3(c)), None, None, 0, 0)

    def texture_checker(self, c1, c2):
        return self.L.orc_add_texture(self.h, TEX_CHECKER, _opt(_f3(c1)), _opt(_f3(c2)), None, 0, 0)

    def texture_image(self, rgba):
        rgba = np.ascontiguousarray(rgba, np.uint8)
        return self.L.orc_add_texture(self.h, TEX_IMAGE, None, None, _opt(rgba), rgba.shape[1], rgba.shape[0])

    def material(self, mtype, tex=-1, emit=None):
        return self.L.orc_add_material(self.h, mtype, tex, None if emit is None else _opt(_f3(emit)))

    def sphere(self, c, r, mat):
        return self.L.orc_add_sphere(self.h, _f3(c), r, mat)

    def cube_minmax(self, mn, mx, mat):
        return self.L.orc_add_cube_minmax(self.h, _f3(mn), _f3(mx), mat)

    def cube_center(self, c, half, mat):
        return self.L.orc_add_cube_center(self.h, _f3(c), half, mat)

    def mesh(self, verts8, size, mat, rot_axis=-1, degrees=0.0, offset=None):
        verts8 = np.ascontiguousarray(verts8, np.float32).reshape(-1, 8)
        off = _f3(offset if offset is not None else (0, 0, 0))
        return self.L.orc_add_mesh(self.h, verts8, len(verts8) // 3, size, mat, rot_axis, degrees,
                                   0 if offset is None else 1, off)

    def fog(self, c, radius, density, mat, rot_axis=-1, degrees=0.0, offset=None):
        off = _f3(offset if offset is not None else (0, 0, 0))
        return self.L.orc_add_fog(self.h, _f3(c), radius, density, mat, rot_axis, degrees, 0 if offset is None else 1, off)

    def plane(self, y, mat):
        return self.L.orc_add_plane(self.h, y, mat)

    def skybox(self, rgb):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        self.L.orc_set_skybox(self.h, rgb, rgb.shape[1], rgb.shape[0])

    def default_scene(self, tree_verts8, tree_rgba, sky_rgb):
        tree_verts8 = np.ascontiguousarray(tree_verts8, np.float32).reshape(-1, 8)
        tree_rgba = np.ascontiguousarray(tree_rgba, np.uint8)
        sky_rgb = np.ascontiguousarray(sky_rgb, np.uint8)
        self.L.orc_default_scene(self.h, tree_verts8, len(tree_verts8) // 3, tree_rgba, tree_rgba.shape[1],
                                 tree_rgba.shape[0], sky_rgb, sky_rgb.shape[1], sky_rgb.shape[0])

    def build(self, build_key):
        self.L.orc_build(self.h, build_key)

    # -- dumps -----------------------------------------------------------------------------------
    def dump_top(self, max_nodes=1 << 20):
        boxes = np.zeros((max_nodes, 6), np.float32)
        left = np.zeros(max_nodes, np.int32)
        right = np.zeros(max_nodes, np.int32)
        leaf = np.zeros(max_nodes + 1, np.int32)
        nl = np.zeros(1, np.int32)
        n = self.L.orc_dump_top(self.h, max_nodes, boxes, left, right, leaf, nl)
        assert n > 0
        return boxes[:n].copy(), left[:n].copy(), right[:n].copy(), leaf[:nl[0]].copy()

    def dump_mesh(self, oid, max_nodes=1 << 16, max_tris=1 << 16):
        boxes = np.zeros((max_nodes, 6), np.float32)
        left = np.zeros(max_nodes, np.int32)
        right = np.zeros(max_nodes, np.int32)
        tris = np.zeros((max_tris, 24), np.float32)
        nt = np.zeros(1, np.int32)
        n = self.L.orc_dump_mesh(self.h, oid, max_nodes, boxes, left, right, max_tris, tris, nt)
        assert n > 0
        return boxes[:n].copy(), left[:n].copy(), right[:n].copy(), tris[:nt[0]].copy()

    def dump_objects(self, max_obj):
        out = np.zeros((max_obj, 20), np.float32)
        n = self.L.orc_dump_objects(self.h, max_obj, out)
        assert n >= 0
        return out[:n].copy()

    def dump_instance(self, oid):
        out = np.zeros(17, np.float32)
        assert self.L.orc_dump_instance(self.h, oid, out) == 1
        return out

    # -- queries ---------------------------------------------------------------------------------
    def trace(self, o, d):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        hit = np.zeros(len(o), np.int32)
        rec = np.zeros((len(o), 9), np.float32)
        self.L.orc_trace(self.h, len(o), o, d, hit, rec)
        return hit, rec

    def ray_color(self, o, d, bounces, keys, sky_brightness=0.5):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        rgb = np.zeros((len(o), 3), np.float32)
        self.L.orc_ray_color(self.h, len(o), o, d, bounces, np.ascontiguousarray(keys, np.uint32), sky_brightness, rgb)
        return rgb

    def skybox_lookup(self, d):
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        rgb = np.zeros((len(d), 3), np.float32)
        self.L.orc_skybox(self.h, len(d), d, rgb)
        return rgb

    def render(self, cam12, W, H, S, B, seed, rect=None, fov=1.5, dof=0.05, focus=10.0, sky=0.5, threads=1,
               want=("radiance", "primary_id", "primary_t", "rgba")):
        x0, y0, x1, y1 = rect if rect else (0, 0, W, H)
        npx = (x1 - x0) * (y1 - y0)
        spp = S * S
        out = {}
        if "radiance" in want:
            out["radiance"] = np.zeros((npx, spp, 3), np.float32)
        if "primary_id" in want:
            out["primary_id"] = np.zeros((npx, spp), np.int32)
        if "primary_t" in want:
            out["primary_t"] = np.zeros((npx, spp), np.float32)
        if "ray" in want:
            out["ray"] = np.zeros((npx, spp, 6), np.float32)
        if "rng_draws" in want:
            out["rng_draws"] = np.zeros((npx, spp), np.uint32)
        if "rgba" in want:
            out["rgba"] = np.zeros((y1 - y0, x1 - x0, 4), np.uint8)
        self.L.orc_render(self.h, np.array([W, H, S, B], np.int32), np.array([fov, dof, focus, sky], np.float32),
                          np.ascontiguousarray(cam12, np.float32).reshape(12), seed, x0, y0, x1, y1, threads,
                          _opt(out.get("radiance")), _opt(out.get("primary_id")), _opt(out.get("primary_t")),
                          _opt(out.get("ray")), _opt(out.get("rng_draws")), _opt(out.get("rgba")))
        return out

    def render_pixel_keyed(self, cam12, W, H, S, B, seed, rect, fov=1.5, dof=0.05, focus=10.0, sky=0.5):
        x0, y0, x1, y1 = rect
        rgba = np.zeros((y1 - y0, x1 - x0, 4), np.uint8)
        self.L.orc_render_pixel_keyed(self.h, np.array([W, H, S, B], np.int32), np.array([fov, dof, focus, sky], np.float32),
                                      np.ascontiguousarray(cam12, np.float32).reshape(12), seed, x0, y0, x1, y1,
                                      rgba.ctypes.data_as(C.c_void_p))
        return rgba

    def render_framemanager(self, cam12, W, H, S, B, n_threads=7, spread=37, fov=1.5, dof=0.05, focus=10.0, sky=0.5):
        rgba = np.zeros((H, W, 4), np.uint8)
        self.L.orc_render_framemanager(self.h, np.array([W, H, S, B], np.int32), np.array([fov, dof, focus, sky], np.float32),
                                       np.ascontiguousarray(cam12, np.float32).reshape(12), n_threads, spread,
                                       rgba.ctypes.data_as(C.c_void_p))
        return rgba

    def counters(self, reset=False):
        out = np.zeros(16, np.uint64)
        self.L.orc_counters(self.h, out, int(reset))
        return dict(zip(COUNTER_NAMES, (int(v) for v in out)))

    def repeat_counters(self):
        """the part of counters() spent on second visits of leaves listed as both children of a node (Object.cpp:198-200)"""
        out = np.zeros(16, np.uint64)
        self.L.orc_repeat_counters(self.h, out)
        return dict(zip(COUNTER_NAMES, (int(v) for v in out)))


def camera(pos, look):
    out = np.zeros(12, np.float32)
    lib().orc_camera(_f3(pos), _f3(look), out)
    return out


def camera_input(cam12, xrel=0.0, yrel=0.0, w=0, a=0, s=0, d=0):
    cam = np.ascontiguousarray(cam12, np.float32).reshape(12).copy()
    lib().orc_camera_input(cam, xrel, yrel, w, a, s, d)
    return cam


def aabb_hit(box6, o, d):
    o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
    out = np.zeros(len(o), np.uint8)
    lib().orc_aabb_hit(np.asarray(box6, np.float32), len(o), o, d, out)
    return out


def normalize(v):
    v = np.ascontiguousarray(v, np.float32).reshape(-1, 3)
    out = np.zeros_like(v)
    lib().orc_normalize(len(v), v, out)
    return out


def intersect_prim(kind, params, o, d, fill=0.0):
    o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
    hit = np.zeros(len(o), np.int32)
    rec = np.zeros((len(o), 9), np.float32)
    p = np.zeros(8, np.float32)
    p[:len(params)] = params
    lib().orc_intersect_prim(kind, p, len(o), o, d, fill, hit, rec)
    return hit, rec


def intersect_tri(verts24, rgba, o, d, fill=0.0):
    o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
    hit = np.zeros(len(o), np.int32)
    rec = np.zeros((len(o), 9), np.float32)
    if rgba is not None:
        rgba = np.ascontiguousarray(rgba, np.uint8)
        lib().orc_intersect_tri(np.ascontiguousarray(verts24, np.float32).reshape(24), _opt(rgba), rgba.shape[1], rgba.shape[0],
                                len(o), o, d, fill, hit, rec)
    else:
        lib().orc_intersect_tri(np.ascontiguousarray(verts24, np.float32).reshape(24), None, 0, 0, len(o), o, d, fill, hit, rec)
    return hit, rec


def std_sort(keys, order=None):
    keys = np.ascontiguousarray(keys, np.float32)
    order = np.arange(len(keys), dtype=np.int32) if order is None else np.ascontiguousarray(order, np.int32).copy()
    lib().orc_std_sort(len(order), keys, order)
    return order


def thread_pixel_order(W, H, n_threads, spread, thread):
    n = lib().orc_thread_pixel_order(W, H, n_threads, spread, thread, 0, np.zeros(1, np.uint32))
    out = np.zeros(n, np.uint32)
    lib().orc_thread_pixel_order(W, H, n_threads, spread, thread, n, out)
    return out
