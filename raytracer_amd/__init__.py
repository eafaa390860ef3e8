"""MI355X-native renderer for the per-pixel path of RaskiTech/Raytracer.

`raytracer_amd.engine` binds the C ABI (include/rt_engine.h) of librt_engine.so: scene construction, camera and the
HIP engine.  `raytracer_amd.distributed` splits a frame across the GPUs of a node by image tiles.
"""
from . import engine  # noqa: F401
from .engine import Config, Engine, MultiEngine, RtError, Scene, alt_scene, camera_input, camera_look, default_config, default_scene  # noqa: F401
