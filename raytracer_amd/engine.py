"""ctypes binding of librt_engine.so (include/rt_engine.h): scene construction, camera, engine.

The library is the product: if it is missing, or if no GPU is present when an engine is created, this module raises —
there is no CPU fallback.  Host-only entry points (scene building, asset decoding, camera) work without a GPU.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# RT_ENGINE_LIB names another build of the same library (A/B measurements of kernel variants)
LIB_PATH = os.environ.get("RT_ENGINE_LIB") or os.path.join(HERE, "librt_engine.so")
ASSET_DIR = os.path.join(os.path.dirname(HERE), "assets")

MAT_NONE, MAT_DIFFUSE, MAT_METAL, MAT_ISOTROPIC, MAT_LIGHT = 0, 1, 2, 3, 4
TEX_COLOR, TEX_CHECKER, TEX_IMAGE = 0, 1, 2
TRACE_STATIC_MAX_RAYS = 32768   # RT_TRACE_STATIC_MAX_RAYS
FLAG_NO_CULL = 1                # RT_FLAG_NO_CULL
FLAG_NO_LDS = 2                 # RT_FLAG_NO_LDS
FLAG_NO_ORDER = 4               # RT_FLAG_NO_ORDER
FLAG_NO_REGROUP = 8             # RT_FLAG_NO_REGROUP
FLAG_NO_PAIRS = 16              # RT_FLAG_NO_PAIRS
FLAG_NO_WIDE = 32               # RT_FLAG_NO_WIDE

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


class RtError(RuntimeError):
    pass


class Config(C.Structure):
    """rt_config — the quality settings of the reference's Constants.h as run-time values."""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("samples_per_axis", C.c_int32), ("bounces", C.c_int32),
                ("fov", C.c_float), ("skybox_brightness", C.c_float), ("dof_intensity", C.c_float), ("focus_distance", C.c_float),
                ("device", C.c_int32), ("max_batch_paths", C.c_int32), ("tile_w", C.c_int32), ("tile_h", C.c_int32),
                ("flags", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays", C.c_uint64), ("rays_at_depth", C.c_uint64 * 33), ("box_tests", C.c_uint64)]


# every symbol include/rt_engine.h declares: name -> (restype, argtypes)
_V = C.c_void_p
_PV = C.POINTER(C.c_void_p)
SIGNATURES = {
    "rt_last_error": (C.c_char_p, []),
    "rt_source_hash": (C.c_char_p, []),
    "rt_config_defaults": (None, [C.POINTER(Config)]),
    "rt_scene_create": (C.c_int, [_PV]),
    "rt_scene_destroy": (None, [_V]),
    "rt_scene_add_texture_color": (C.c_int, [_V, f32p]),
    "rt_scene_add_texture_checker": (C.c_int, [_V, f32p, f32p]),
    "rt_scene_add_texture_image": (C.c_int, [_V, u8p, C.c_int, C.c_int]),
    "rt_scene_add_texture_file": (C.c_int, [_V, C.c_char_p]),
    "rt_scene_add_material": (C.c_int, [_V, C.c_int, C.c_int, _V]),
    "rt_scene_add_sphere": (C.c_int, [_V, f32p, C.c_float, C.c_int]),
    "rt_scene_add_cube_center": (C.c_int, [_V, f32p, C.c_float, C.c_int]),
    "rt_scene_add_cube_minmax": (C.c_int, [_V, f32p, f32p, C.c_int]),
    "rt_scene_add_plane": (C.c_int, [_V, C.c_float, C.c_int]),
    "rt_scene_add_mesh": (C.c_int, [_V, f32p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_float, C.c_int, f32p]),
    "rt_scene_add_fog": (C.c_int, [_V, f32p, C.c_float, C.c_float, C.c_int, C.c_int, C.c_float, C.c_int, f32p]),
    "rt_scene_add_mesh_file": (C.c_int, [_V, C.c_char_p, C.c_float, C.c_int, C.c_int, C.c_float, C.c_int, f32p]),
    "rt_scene_set_skybox": (C.c_int, [_V, u8p, C.c_int, C.c_int]),
    "rt_scene_set_skybox_file": (C.c_int, [_V, C.c_char_p]),
    "rt_scene_add_default_objects": (C.c_int, [_V, C.c_char_p, C.c_float]),
    "rt_scene_add_alt_objects": (C.c_int, [_V, C.c_char_p, C.c_float]),
    "rt_scene_build": (C.c_int, [_V, C.c_uint32]),
    "rt_scene_dump_top": (C.c_int, [_V, C.c_int, f32p, i32p, i32p, i32p, i32p]),
    "rt_scene_dump_mesh": (C.c_int, [_V, C.c_int, C.c_int, f32p, i32p, i32p, C.c_int, f32p, i32p]),
    "rt_scene_dump_objects": (C.c_int, [_V, C.c_int, f32p]),
    "rt_scene_dump_instance": (C.c_int, [_V, C.c_int, f32p]),
    "rt_scene_flat_info": (C.c_int, [_V, i64p]),
    "rt_scene_regroup_info": (C.c_int, [_V, f64p]),
    "rt_scene_wide_info": (C.c_int, [_V, f64p]),
    "rt_scene_dump_entries": (C.c_int, [_V, C.c_int, C.c_int, _V, _V, _V]),
    "rt_load_obj": (C.c_int, [C.c_char_p, C.c_int, _V]),
    "rt_load_png": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int64, _V, i32p, i32p, i32p]),
    "rt_camera_look": (None, [f32p, f32p, f32p]),
    "rt_camera_input": (None, [f32p, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int]),
    "rt_engine_create": (C.c_int, [C.POINTER(Config), _PV]),
    "rt_engine_destroy": (None, [_V]),
    "rt_engine_set_scene": (C.c_int, [_V, _V]),
    "rt_engine_set_camera": (C.c_int, [_V, f32p]),
    "rt_engine_get_config": (C.c_int, [_V, C.POINTER(Config)]),
    "rt_engine_render": (C.c_int, [_V, C.c_uint32, C.c_int, C.c_int, _V, _V]),
    "rt_engine_tile_bytes": (C.c_int64, [_V, C.c_int]),
    "rt_engine_set_window": (C.c_int, [_V, C.c_int, C.c_int, C.c_int, C.c_int]),
    "rt_engine_assemble": (C.c_int, [_V, _V, C.c_int, _V, _V]),
    "rt_engine_sync": (C.c_int, [_V]),
    "rt_engine_poll": (C.c_int, [_V]),
    "rt_engine_read_frame": (C.c_int, [_V, u8p]),
    "rt_engine_read_rect": (C.c_int, [_V, C.c_int, C.c_int, C.c_int, C.c_int, u8p]),
    "rt_engine_map_frame": (C.c_int, [_V, C.c_int, _PV]),
    "rt_engine_clear_frame": (C.c_int, [_V, _V]),
    "rt_engine_cancel": (C.c_int, [_V]),
    "rt_engine_keep_samples": (C.c_int, [_V, C.c_int]),
    "rt_engine_read_samples": (C.c_int, [_V, _V, _V, _V, _V]),
    "rt_engine_trace": (C.c_int, [_V, C.c_int, f32p, f32p, i32p, f32p]),
    "rt_engine_trace_static": (C.c_int, [_V, C.c_int, f32p, f32p, i32p, f32p]),
    "rt_engine_ray_color": (C.c_int, [_V, C.c_int, f32p, f32p, C.c_int, u32p, f32p]),
    "rt_engine_get_stats": (C.c_int, [_V, C.POINTER(Stats)]),
    "rt_engine_time_kernels": (C.c_int, [_V, C.c_int]),
    "rt_engine_kernel_times": (C.c_int, [_V, f64p]),
    "rt_multi_create": (C.c_int, [C.POINTER(Config), i32p, C.c_int, _PV]),
    "rt_multi_destroy": (None, [_V]),
    "rt_multi_count": (C.c_int, [_V]),
    "rt_multi_engine": (C.c_void_p, [_V, C.c_int]),
    "rt_multi_set_scene": (C.c_int, [_V, _V]),
    "rt_multi_set_camera": (C.c_int, [_V, f32p]),
    "rt_multi_render": (C.c_int, [_V, C.c_uint32]),
    "rt_multi_poll": (C.c_int, [_V]),
    "rt_multi_sync": (C.c_int, [_V]),
    "rt_multi_cancel": (C.c_int, [_V]),
    "rt_multi_clear_frame": (C.c_int, [_V]),
    "rt_multi_map_frame": (C.c_int, [_V, C.c_int, _PV]),
    "rt_engine_eval_math": (C.c_int, [_V, C.c_int, f32p, f32p, f32p, f32p]),
    "rt_engine_eval_rng": (C.c_int, [_V, C.c_int, u32p, u32p, u32p, f32p]),
    "rt_engine_check_division": (C.c_int, [_V, C.c_int, f32p, f32p, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_uint64), f32p]),
}

_L = None


def lib():
    """Load librt_engine.so and bind every declared entry point (raises if the library or a symbol is missing)."""
    global _L
    if _L is not None:
        return _L
    if not os.path.exists(LIB_PATH):
        raise RtError(f"{LIB_PATH} is missing: build it with `make -C raytracer_amd/csrc` (or __graft_entry__.build())")
    # PyTorch ships its own libamdhip64 (soname libamdhip64.so.7).  A process must hold ONE HIP runtime, or whichever
    # loads second sees no GPU: import torch first, so that this library's libamdhip64.so.7 dependency resolves to the
    # copy torch already mapped and device pointers / streams can be shared with torch tensors and RCCL.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _L = L
    return L


def source_hash():
    """digest of the kernel sources the loaded library was compiled from"""
    return lib().rt_source_hash().decode()


def _check(rc):
    if rc < 0:
        raise RtError(lib().rt_last_error().decode() or f"rt error {rc}")
    return rc


def _f3(v):
    return np.ascontiguousarray(v, np.float32).reshape(3)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def default_config(**kw):
    cfg = Config()
    lib().rt_config_defaults(C.byref(cfg))
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


def camera_look(pos, look):
    """Camera(pos, look) of the reference (DataUtility.h:137): returns cam12 = pos, forward, u, v."""
    out = np.zeros(12, np.float32)
    lib().rt_camera_look(_f3(pos), _f3(look), out)
    return out


def camera_input(cam12, xrel=0.0, yrel=0.0, w=0, a=0, s=0, d=0):
    cam = np.ascontiguousarray(cam12, np.float32).reshape(12).copy()
    lib().rt_camera_input(cam, xrel, yrel, w, a, s, d)
    return cam


def load_obj(path, max_verts=1 << 22):
    n = _check(lib().rt_load_obj(os.fsencode(path), max_verts, None))
    out = np.zeros((n, 8), np.float32)
    _check(lib().rt_load_obj(os.fsencode(path), n, _ptr(out)))
    return out


def load_png(path, channels, flip=True):
    w = np.zeros(1, np.int32)
    h = np.zeros(1, np.int32)
    fc = np.zeros(1, np.int32)
    n = _check(lib().rt_load_png(os.fsencode(path), channels, int(flip), 1 << 40, None, w, h, fc))
    out = np.zeros(n, np.uint8)
    _check(lib().rt_load_png(os.fsencode(path), channels, int(flip), n, _ptr(out), w, h, fc))
    return out.reshape(h[0], w[0], channels if channels > 0 else fc[0]), int(fc[0])


class Scene:
    """Host-side scene: objects in creation order get ids 0,1,2,...; planes continue the numbering."""

    def __init__(self):
        self.L = lib()
        h = C.c_void_p()
        _check(self.L.rt_scene_create(C.byref(h)))
        self.h = h

    def __del__(self):
        if getattr(self, "h", None):
            self.L.rt_scene_destroy(self.h)
            self.h = None

    def texture_color(self, rgb):
        return _check(self.L.rt_scene_add_texture_color(self.h, _f3(rgb)))

    def texture_checker(self, a, b):
        return _check(self.L.rt_scene_add_texture_checker(self.h, _f3(a), _f3(b)))

    def texture_image(self, rgba):
        rgba = np.ascontiguousarray(rgba, np.uint8)
        return _check(self.L.rt_scene_add_texture_image(self.h, rgba, rgba.shape[1], rgba.shape[0]))

    def texture_file(self, path):
        return _check(self.L.rt_scene_add_texture_file(self.h, os.fsencode(path)))

    def material(self, mtype, tex=-1, emit=None):
        e = None if emit is None else _f3(emit)
        return _check(self.L.rt_scene_add_material(self.h, mtype, tex, _ptr(e)))

    def sphere(self, c, r, mat):
        return _check(self.L.rt_scene_add_sphere(self.h, _f3(c), r, mat))

    def cube_center(self, c, half, mat):
        return _check(self.L.rt_scene_add_cube_center(self.h, _f3(c), half, mat))

    def cube_minmax(self, mn, mx, mat):
        return _check(self.L.rt_scene_add_cube_minmax(self.h, _f3(mn), _f3(mx), mat))

    def plane(self, y, mat):
        return _check(self.L.rt_scene_add_plane(self.h, y, mat))

    def mesh(self, verts8, size, mat, rot_axis=-1, degrees=0.0, offset=None):
        verts8 = np.ascontiguousarray(verts8, np.float32).reshape(-1, 8)
        off = _f3(offset if offset is not None else (0, 0, 0))
        return _check(self.L.rt_scene_add_mesh(self.h, verts8, len(verts8) // 3, size, mat, rot_axis, degrees,
                                               0 if offset is None else 1, off))

    def fog(self, c, radius, density, mat, rot_axis=-1, degrees=0.0, offset=None):
        off = _f3(offset if offset is not None else (0, 0, 0))
        return _check(self.L.rt_scene_add_fog(self.h, _f3(c), radius, density, mat, rot_axis, degrees, 0 if offset is None else 1, off))

    def mesh_file(self, path, size, mat, rot_axis=-1, degrees=0.0, offset=None):
        off = _f3(offset if offset is not None else (0, 0, 0))
        return _check(self.L.rt_scene_add_mesh_file(self.h, os.fsencode(path), size, mat, rot_axis, degrees,
                                                    0 if offset is None else 1, off))

    def skybox(self, rgb):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        _check(self.L.rt_scene_set_skybox(self.h, rgb, rgb.shape[1], rgb.shape[0]))

    def skybox_file(self, path):
        _check(self.L.rt_scene_set_skybox_file(self.h, os.fsencode(path)))

    def default_objects(self, asset_dir=ASSET_DIR, time=0.0):
        """The scene the reference compiles in (World.cpp:15-32)."""
        _check(self.L.rt_scene_add_default_objects(self.h, os.fsencode(asset_dir), time))

    def build(self, build_key):
        _check(self.L.rt_scene_build(self.h, build_key))
        return self

    # -- introspection ---------------------------------------------------------------------------------------------
    def dump_top(self, max_nodes=1 << 20):
        boxes = np.zeros((max_nodes, 6), np.float32)
        left = np.zeros(max_nodes, np.int32)
        right = np.zeros(max_nodes, np.int32)
        leaf = np.zeros(max_nodes + 1, np.int32)
        nl = np.zeros(1, np.int32)
        n = _check(self.L.rt_scene_dump_top(self.h, max_nodes, boxes, left, right, leaf, nl))
        return boxes[:n].copy(), left[:n].copy(), right[:n].copy(), leaf[:nl[0]].copy()

    def dump_mesh(self, oid, max_nodes=1 << 16, max_tris=1 << 16):
        boxes = np.zeros((max_nodes, 6), np.float32)
        left = np.zeros(max_nodes, np.int32)
        right = np.zeros(max_nodes, np.int32)
        tris = np.zeros((max_tris, 24), np.float32)
        nt = np.zeros(1, np.int32)
        n = _check(self.L.rt_scene_dump_mesh(self.h, oid, max_nodes, boxes, left, right, max_tris, tris, nt))
        return boxes[:n].copy(), left[:n].copy(), right[:n].copy(), tris[:nt[0]].copy()

    def dump_objects(self, max_obj):
        out = np.zeros((max_obj, 20), np.float32)
        n = _check(self.L.rt_scene_dump_objects(self.h, max_obj, out))
        return out[:n].copy()

    def dump_instance(self, oid):
        out = np.zeros(17, np.float32)
        _check(self.L.rt_scene_dump_instance(self.h, oid, out))
        return out

    def dump_entries(self, regrouped=False):
        """(lo_hi [n, 8] float32, peer [n] int32, rank [n] uint32) of the reference's layout, the regrouped one (True / 1) or
        the regrouped one with wide nodes (2)"""
        n = _check(self.L.rt_scene_dump_entries(self.h, int(regrouped), 0, None, None, None))
        e = np.zeros((n, 8), np.float32)
        peer = np.zeros(n, np.int32)
        rank = np.zeros(n, np.uint32)
        _check(self.L.rt_scene_dump_entries(self.h, int(regrouped), n, _ptr(e), _ptr(peer), _ptr(rank)))
        return e, peer, rank

    def wide_info(self):
        out = np.zeros(4, np.float64)
        _check(self.L.rt_scene_wide_info(self.h, out))
        return dict(zip(("has_wide_layout", "entries", "mesh_tests_wide", "most_children"), (float(v) for v in out)))

    def regroup_info(self):
        out = np.zeros(8, np.float64)
        _check(self.L.rt_scene_regroup_info(self.h, out))
        return dict(zip(("has_regrouped_layout", "trees", "lowest_nodes", "top_tests_reference", "mesh_tests_reference", "top_tests_regrouped",
                         "mesh_tests_regrouped", "stack_slots"), (float(v) for v in out)))

    def flat_info(self):
        out = np.zeros(7, np.int64)
        _check(self.L.rt_scene_flat_info(self.h, out))
        return dict(zip(("entries", "triangles", "instances", "objects", "planes", "texel_bytes", "device_bytes"), (int(v) for v in out)))


def alt_scene(build_key, time=0.0, asset_dir=ASSET_DIR):
    """the scene of World.cpp:34-40: room, moving light, rotated fog volume"""
    s = Scene()
    _check(s.L.rt_scene_add_alt_objects(s.h, os.fsencode(asset_dir), time))
    return s.build(build_key)


def default_scene(build_key, asset_dir=ASSET_DIR):
    s = Scene()
    s.default_objects(asset_dir)
    return s.build(build_key)


class Engine:
    """One GPU's renderer (rt_engine).  Requires a HIP device."""

    def __init__(self, cfg=None, **kw):
        self.L = lib()
        self.cfg = cfg if cfg is not None else default_config(**kw)
        h = C.c_void_p()
        _check(self.L.rt_engine_create(C.byref(self.cfg), C.byref(h)))
        self.h = h
        got = Config()
        _check(self.L.rt_engine_get_config(self.h, C.byref(got)))
        self.cfg = got
        self._scene = None
        self.window = (0, 0, self.cfg.width, self.cfg.height)

    def __del__(self):
        self.close()

    def close(self):
        if getattr(self, "h", None):
            self.L.rt_engine_destroy(self.h)
            self.h = None

    @property
    def spp(self):
        return self.cfg.samples_per_axis ** 2

    def set_scene(self, scene):
        _check(self.L.rt_engine_set_scene(self.h, scene.h))
        self._scene = scene

    def set_camera(self, cam12):
        _check(self.L.rt_engine_set_camera(self.h, np.ascontiguousarray(cam12, np.float32).reshape(12)))

    def render(self, seed, rank=0, n_ranks=1, d_out=None, stream=None):
        """Asynchronous.  d_out: device pointer (int) or None for the engine's own frame buffer."""
        _check(self.L.rt_engine_render(self.h, seed, rank, n_ranks, d_out, stream))

    def set_window(self, rect=None):
        """rect = (x0, y0, x1, y1) or None for the whole frame"""
        x0, y0, x1, y1 = rect if rect else (0, 0, self.cfg.width, self.cfg.height)
        _check(self.L.rt_engine_set_window(self.h, x0, y0, x1, y1))
        self.window = (x0, y0, x1, y1)

    def tile_bytes(self, n_ranks):
        return int(self.L.rt_engine_tile_bytes(self.h, n_ranks))

    def assemble(self, d_gathered, n_ranks, d_out=None, stream=None):
        _check(self.L.rt_engine_assemble(self.h, d_gathered, n_ranks, d_out, stream))

    def sync(self):
        _check(self.L.rt_engine_sync(self.h))

    def poll(self):
        """True when everything queued on the engine's stream has finished (never blocks)."""
        return bool(_check(self.L.rt_engine_poll(self.h)))

    def read_frame(self):
        out = np.zeros((self.cfg.height, self.cfg.width, 4), np.uint8)
        _check(self.L.rt_engine_read_frame(self.h, out))
        return out

    def read_rect(self, x0, y0, x1, y1):
        """the pixels [x0,x1) x [y0,y1) of the engine's own frame buffer"""
        out = np.zeros((y1 - y0, x1 - x0, 4), np.uint8)
        _check(self.L.rt_engine_read_rect(self.h, x0, y0, x1, y1, out))
        return out

    def render_frame(self, seed):
        self.render(seed)
        return self.read_frame()

    def map_frame(self, wait=True):
        """The frame in the engine's page-locked host buffer, as a read-only view (no copy); valid until the next call.
        wait=False: a snapshot of the frame buffer while the render goes on."""
        p = C.c_void_p()
        _check(self.L.rt_engine_map_frame(self.h, int(wait), C.byref(p)))
        n = self.cfg.height * self.cfg.width * 4
        buf = (C.c_uint8 * n).from_address(p.value)
        a = np.frombuffer(buf, np.uint8).reshape(self.cfg.height, self.cfg.width, 4)
        a.flags.writeable = False
        return a

    def clear_frame(self, stream=None):
        _check(self.L.rt_engine_clear_frame(self.h, stream))

    def cancel(self):
        """The render in flight stops taking rays (returns at once; sync() then returns early)."""
        _check(self.L.rt_engine_cancel(self.h))

    def keep_samples(self, enable=True):
        _check(self.L.rt_engine_keep_samples(self.h, int(enable)))

    def read_samples(self, want=("radiance", "primary_id", "primary_t", "rng_draws")):
        n = (self.window[2] - self.window[0]) * (self.window[3] - self.window[1])
        out = {}
        if "radiance" in want:
            out["radiance"] = np.zeros((n, self.spp, 3), np.float32)
        if "primary_id" in want:
            out["primary_id"] = np.zeros((n, self.spp), np.int32)
        if "primary_t" in want:
            out["primary_t"] = np.zeros((n, self.spp), np.float32)
        if "rng_draws" in want:
            out["rng_draws"] = np.zeros((n, self.spp), np.uint32)
        _check(self.L.rt_engine_read_samples(self.h, _ptr(out.get("radiance")), _ptr(out.get("primary_id")),
                                             _ptr(out.get("primary_t")), _ptr(out.get("rng_draws"))))
        return out

    def trace(self, o, d):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        hit = np.zeros(len(o), np.int32)
        rec = np.zeros((len(o), 9), np.float32)
        _check(self.L.rt_engine_trace(self.h, len(o), o, d, hit, rec))
        return hit, rec

    def trace_static(self, o, d):
        """trace() through the entry point that allocates nothing (at most TRACE_STATIC_MAX_RAYS rays)"""
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        hit = np.zeros(len(o), np.int32)
        rec = np.zeros((len(o), 9), np.float32)
        _check(self.L.rt_engine_trace_static(self.h, len(o), o, d, hit, rec))
        return hit, rec

    def ray_color(self, o, d, bounces, keys):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        rgb = np.zeros((len(o), 3), np.float32)
        _check(self.L.rt_engine_ray_color(self.h, len(o), o, d, bounces, np.ascontiguousarray(keys, np.uint32), rgb))
        return rgb

    def stats(self):
        st = Stats()
        _check(self.L.rt_engine_get_stats(self.h, C.byref(st)))
        return {"paths": int(st.paths), "rays": int(st.rays), "rays_at_depth": [int(v) for v in st.rays_at_depth],
                "box_tests": int(st.box_tests)}

    def time_kernels(self, enable=True):
        _check(self.L.rt_engine_time_kernels(self.h, int(enable)))

    def kernel_times(self):
        out = np.zeros(2, np.float64)
        _check(self.L.rt_engine_kernel_times(self.h, out))
        return float(out[0]), int(out[1])

    def eval_math(self, x, y):
        x = np.ascontiguousarray(x, np.float32)
        y = np.ascontiguousarray(y, np.float32)
        s = np.zeros_like(x)
        a = np.zeros_like(x)
        _check(self.L.rt_engine_eval_math(self.h, len(x), x, y, s, a))
        return s, a

    def check_division(self, a, b, seed=1, threads=1 << 20, rounds=256):
        """(mismatching pairs, first mismatch) of the reciprocal-based division vs IEEE division on the GPU"""
        a = np.ascontiguousarray(a, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        m = C.c_uint64(0)
        bad = np.zeros(4, np.float32)
        _check(self.L.rt_engine_check_division(self.h, len(a), a, b, seed, threads, rounds, C.byref(m), bad))
        return int(m.value), bad

    def eval_rng(self, key, ctr):
        key = np.ascontiguousarray(key, np.uint32)
        ctr = np.ascontiguousarray(ctr, np.uint32)
        r = np.zeros(len(key), np.uint32)
        u = np.zeros(len(key), np.float32)
        _check(self.L.rt_engine_eval_rng(self.h, len(key), key, ctr, r, u))
        return r, u


class MultiEngine:
    """Several GPUs behind one handle in one process (rt_multi): device k renders tiles k, k+n, ...; the first device
    gathers them with peer-to-peer copies and holds the frame.  A device may be listed more than once."""

    def __init__(self, cfg, device_ids):
        self.L = lib()
        ids = np.ascontiguousarray(device_ids, np.int32)
        h = C.c_void_p()
        _check(self.L.rt_multi_create(C.byref(cfg), ids, len(ids), C.byref(h)))
        self.h = h
        self.cfg = Config()
        _check(self.L.rt_engine_get_config(self.L.rt_multi_engine(self.h, 0), C.byref(self.cfg)))
        self._scene = None

    def __del__(self):
        self.close()

    def close(self):
        if getattr(self, "h", None):
            self.L.rt_multi_destroy(self.h)
            self.h = None

    def __len__(self):
        return int(self.L.rt_multi_count(self.h))

    def set_scene(self, scene):
        _check(self.L.rt_multi_set_scene(self.h, scene.h))
        self._scene = scene

    def set_camera(self, cam12):
        _check(self.L.rt_multi_set_camera(self.h, np.ascontiguousarray(cam12, np.float32).reshape(12)))

    def render(self, seed):
        _check(self.L.rt_multi_render(self.h, seed))

    def poll(self):
        return bool(_check(self.L.rt_multi_poll(self.h)))

    def sync(self):
        _check(self.L.rt_multi_sync(self.h))

    def cancel(self):
        _check(self.L.rt_multi_cancel(self.h))

    def clear_frame(self):
        _check(self.L.rt_multi_clear_frame(self.h))

    def map_frame(self, wait=True):
        p = C.c_void_p()
        _check(self.L.rt_multi_map_frame(self.h, int(wait), C.byref(p)))
        n = self.cfg.height * self.cfg.width * 4
        a = np.frombuffer((C.c_uint8 * n).from_address(p.value), np.uint8).reshape(self.cfg.height, self.cfg.width, 4)
        a.flags.writeable = False
        return a

    def stats(self, k):
        st = Stats()
        _check(self.L.rt_engine_get_stats(self.L.rt_multi_engine(self.h, k), C.byref(st)))
        return {"paths": int(st.paths), "rays": int(st.rays), "box_tests": int(st.box_tests)}
