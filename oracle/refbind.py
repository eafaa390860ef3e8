"""TEST INFRASTRUCTURE: ctypes binding of oracle/_ref/libref.so (the reference's own C++ compiled with
oracle/ref_harness.cpp).  Used by oracle/gen_golden.py and, where the library exists, by tests and by
bench.py's cpu_baseline leg.  Never imported by the product package."""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_ref", "libref.so")
# The reference opens "src/stb_image/..." relative to the cwd (World.cpp:26,49); assets/ mirrors that layout.
ASSET_ROOT = os.path.join(os.path.dirname(HERE), "assets")

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def available():
    return os.path.exists(LIB_PATH)


class _Cwd:
    def __enter__(self):
        self.old = os.getcwd()
        os.chdir(ASSET_ROOT)

    def __exit__(self, *a):
        os.chdir(self.old)


def _opt(arr):
    return None if arr is None else arr.ctypes.data_as(C.c_void_p)


class Ref:
    def __init__(self):
        L = C.CDLL(LIB_PATH)
        self.L = L
        L.ref_world_new.restype = C.c_void_p
        L.ref_world_new.argtypes = [C.c_uint32]
        L.ref_world_free.argtypes = [C.c_void_p]
        L.ref_rng_set.argtypes = [C.c_uint32, C.c_uint32]
        L.ref_rng_ctr.restype = C.c_uint32
        L.ref_rng_calls.restype = C.c_uint64
        L.ref_rand31.restype = C.c_uint32
        L.ref_rand31.argtypes = [C.c_uint32, C.c_uint32]
        L.ref_sample_key.restype = C.c_uint32
        L.ref_sample_key.argtypes = [C.c_uint32] * 3
        L.ref_random01.restype = C.c_float
        L.ref_random_unit_distance.restype = C.c_float
        L.ref_unit_sphere_point.argtypes = [f32p]
        L.ref_unit_circle_point.argtypes = [f32p]
        L.ref_world_set_camera.argtypes = [C.c_void_p, f32p, f32p]
        L.ref_world_get_camera.argtypes = [C.c_void_p, f32p]
        L.ref_world_camera_input.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int]
        L.ref_world_set_custom.argtypes = [C.c_void_p, C.c_uint32, C.c_int, f32p, C.c_int, f32p, C.c_int, f32p, C.c_int, f32p]
        L.ref_dump_top.restype = C.c_int
        L.ref_dump_top.argtypes = [C.c_void_p, C.c_int, f32p, i32p, i32p, i32p, i32p]
        L.ref_dump_objects.restype = C.c_int
        L.ref_dump_objects.argtypes = [C.c_void_p, C.c_int, f32p]
        L.ref_dump_instance.restype = C.c_int
        L.ref_dump_instance.argtypes = [C.c_void_p, C.c_int, f32p]
        L.ref_dump_mesh.restype = C.c_int
        L.ref_dump_mesh.argtypes = [C.c_void_p, C.c_int, C.c_int, f32p, i32p, i32p, C.c_int, f32p, i32p]
        L.ref_load_obj.restype = C.c_int
        L.ref_load_obj.argtypes = [C.c_char_p, C.c_int, f32p]
        L.ref_load_image.restype = C.c_int
        L.ref_load_image.argtypes = [C.c_char_p, C.c_int, C.c_int, u8p, i32p, i32p, i32p]
        L.ref_skybox_info.argtypes = [C.c_void_p, i32p]
        L.ref_aabb_hit.argtypes = [f32p, C.c_int, f32p, f32p, u8p]
        L.ref_normalize.argtypes = [C.c_int, f32p, f32p]
        L.ref_intersect_prim.argtypes = [C.c_int, f32p, C.c_int, f32p, f32p, C.c_float, i32p, f32p]
        L.ref_intersect_tri.argtypes = [f32p, C.c_int, C.c_int, f32p, f32p, C.c_float, i32p, f32p]
        L.ref_trace.argtypes = [C.c_void_p, C.c_int, f32p, f32p, i32p, f32p]
        L.ref_ray_color.argtypes = [C.c_void_p, C.c_int, f32p, f32p, C.c_int, u32p, f32p]
        L.ref_skybox.argtypes = [C.c_void_p, C.c_int, f32p, f32p]
        L.ref_render.argtypes = [C.c_void_p, i32p, f32p, C.c_uint32] + [C.c_int] * 5 + [C.c_void_p] * 6
        L.ref_check_pixel_loop.restype = C.c_long
        L.ref_check_pixel_loop.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.ref_constants.argtypes = [i32p, f32p]
        L.ref_framemanager_frame.restype = C.c_double
        L.ref_framemanager_frame.argtypes = [C.c_uint32, C.c_void_p]

    # -- world ---------------------------------------------------------------------------------
    def world(self, build_key):
        with _Cwd():
            return self.L.ref_world_new(build_key)

    def free(self, w):
        self.L.ref_world_free(w)

    def set_camera(self, w, pos, look):
        self.L.ref_world_set_camera(w, np.asarray(pos, np.float32), np.asarray(look, np.float32))

    def get_camera(self, w):
        out = np.zeros(12, np.float32)
        self.L.ref_world_get_camera(w, out)
        return out.reshape(4, 3)

    def set_custom(self, w, build_key, tex, mat, obj, plane):
        tex = np.ascontiguousarray(tex, np.float32).reshape(-1, 7)
        mat = np.ascontiguousarray(mat, np.float32).reshape(-1, 5)
        obj = np.ascontiguousarray(obj, np.float32).reshape(-1, 10)
        plane = np.ascontiguousarray(plane, np.float32).reshape(-1, 2)
        with _Cwd():
            self.L.ref_world_set_custom(w, build_key, len(tex), tex, len(mat), mat, len(obj), obj, len(plane), plane)

    # -- dumps ---------------------------------------------------------------------------------
    def dump_top(self, w, max_nodes=1 << 20):
        boxes = np.zeros((max_nodes, 6), np.float32)
        left = np.zeros(max_nodes, np.int32)
        right = np.zeros(max_nodes, np.int32)
        leaf = np.zeros(max_nodes + 1, np.int32)
        nl = np.zeros(1, np.int32)
        n = self.L.ref_dump_top(w, max_nodes, boxes, left, right, leaf, nl)
        assert n > 0
        return boxes[:n].copy(), left[:n].copy(), right[:n].copy(), leaf[:nl[0]].copy()

    def dump_objects(self, w, max_obj):
        out = np.zeros((max_obj, 20), np.float32)
        n = self.L.ref_dump_objects(w, max_obj, out)
        assert n >= 0
        return out[:n].copy()

    def dump_instance(self, w, oid):
        out = np.zeros(17, np.float32)
        assert self.L.ref_dump_instance(w, oid, out) == 1
        return out

    def dump_mesh(self, w, oid, max_nodes=1 << 16, max_tris=1 << 16):
        boxes = np.zeros((max_nodes, 6), np.float32)
        left = np.zeros(max_nodes, np.int32)
        right = np.zeros(max_nodes, np.int32)
        tris = np.zeros((max_tris, 24), np.float32)
        nt = np.zeros(1, np.int32)
        n = self.L.ref_dump_mesh(w, oid, max_nodes, boxes, left, right, max_tris, tris, nt)
        assert n > 0
        return boxes[:n].copy(), left[:n].copy(), right[:n].copy(), tris[:nt[0]].copy()

    def load_obj(self, relpath, max_verts=1 << 20):
        out = np.zeros((max_verts, 8), np.float32)
        with _Cwd():
            n = self.L.ref_load_obj(relpath.encode(), max_verts, out)
        assert n >= 0
        return out[:n].copy()

    def load_image(self, relpath, channels):
        w = np.zeros(1, np.int32)
        h = np.zeros(1, np.int32)
        fc = np.zeros(1, np.int32)
        buf = np.zeros(64 << 20, np.uint8)
        with _Cwd():
            n = self.L.ref_load_image(relpath.encode(), channels, buf.size, buf, w, h, fc)
        assert 0 < n <= buf.size
        return buf[:n].reshape(h[0], w[0], channels).copy(), int(fc[0])

    # -- primitive level -------------------------------------------------------------------------
    def aabb_hit(self, box6, o, d):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        out = np.zeros(len(o), np.uint8)
        self.L.ref_aabb_hit(np.asarray(box6, np.float32), len(o), o, d, out)
        return out

    def normalize(self, v):
        v = np.ascontiguousarray(v, np.float32).reshape(-1, 3)
        out = np.zeros_like(v)
        self.L.ref_normalize(len(v), v, out)
        return out

    def intersect_prim(self, kind, params, o, d, fill=0.0):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        hit = np.zeros(len(o), np.int32)
        rec = np.zeros((len(o), 9), np.float32)
        p = np.zeros(8, np.float32)
        p[:len(params)] = params
        self.L.ref_intersect_prim(kind, p, len(o), o, d, fill, hit, rec)
        return hit, rec

    def intersect_tri(self, verts24, use_alpha, o, d, fill=0.0):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        hit = np.zeros(len(o), np.int32)
        rec = np.zeros((len(o), 9), np.float32)
        with _Cwd():
            self.L.ref_intersect_tri(np.ascontiguousarray(verts24, np.float32).reshape(24), int(use_alpha), len(o), o, d, fill, hit, rec)
        return hit, rec

    # -- scene level -----------------------------------------------------------------------------
    def trace(self, w, o, d):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        hit = np.zeros(len(o), np.int32)
        rec = np.zeros((len(o), 9), np.float32)
        self.L.ref_trace(w, len(o), o, d, hit, rec)
        return hit, rec

    def ray_color(self, w, o, d, bounces, keys):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        rgb = np.zeros((len(o), 3), np.float32)
        self.L.ref_ray_color(w, len(o), o, d, bounces, np.ascontiguousarray(keys, np.uint32), rgb)
        return rgb

    def skybox(self, w, d):
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        rgb = np.zeros((len(d), 3), np.float32)
        self.L.ref_skybox(w, len(d), d, rgb)
        return rgb

    def render(self, w, W, H, S, B, seed, rect=None, fov=1.5, dof=0.05, focus=10.0, threads=1,
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
        self.L.ref_render(w, np.array([W, H, S, B], np.int32), np.array([fov, dof, focus], np.float32), seed,
                          x0, y0, x1, y1, threads, _opt(out.get("radiance")), _opt(out.get("primary_id")),
                          _opt(out.get("primary_t")), _opt(out.get("ray")), _opt(out.get("rng_draws")), _opt(out.get("rgba")))
        return out

    def check_pixel_loop(self, w, seed, rect):
        x0, y0, x1, y1 = rect
        rgba = np.zeros((y1 - y0, x1 - x0, 4), np.uint8)
        bad = self.L.ref_check_pixel_loop(w, seed, x0, y0, x1, y1, rgba.ctypes.data_as(C.c_void_p))
        return bad, rgba

    def constants(self):
        i4 = np.zeros(4, np.int32)
        f4 = np.zeros(4, np.float32)
        self.L.ref_constants(i4, f4)
        return i4, f4

    def framemanager_frame(self, build_key):
        i4, _ = self.constants()
        rgba = np.zeros((i4[1], i4[0], 4), np.uint8)
        with _Cwd():
            secs = self.L.ref_framemanager_frame(build_key, rgba.ctypes.data_as(C.c_void_p))
        return secs, rgba
