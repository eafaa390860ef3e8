"""rt_engine_trace_static — the closest-hit query that allocates nothing — against rt_engine_trace, through the C ABI.

Two calls in a row on the same engine, the second with fewer (and different) rays: the static staging array keeps the
first call's contents, and none of them may leak into the second call's results.  Bit for bit: ids, distances (stale
HitInfo.distance values included), points, normals, texture coordinates."""
import numpy as np
import pytest

from conftest import assert_bits

pytestmark = pytest.mark.gpu


def test_trace_static_matches_trace(rt, golden):
    g = golden("trace_default")
    scene = rt.default_scene(int(golden("scene_default")["build_key"]))
    e = rt.Engine(rt.default_config(width=64, height=64, samples_per_axis=1, bounces=1))
    e.set_scene(scene)
    e.set_camera(rt.camera_look((0.0, 3.0, -10.0), (0.0, 0.0, 1.0)))
    o = np.concatenate([g["rays_o"], g["surf_o"]]).astype(np.float32)
    d = np.concatenate([g["rays_d"], g["surf_d"]]).astype(np.float32)
    rng = np.random.default_rng(3)
    big = rng.permutation(len(o))[: min(len(o), rt.engine.TRACE_STATIC_MAX_RAYS)]
    small = rng.permutation(len(o))[:1237]
    for sel, what in ((big, "first call"), (small, "second, smaller call")):
        hit, rec = e.trace(o[sel], d[sel])
        hit_s, rec_s = e.trace_static(o[sel], d[sel])
        assert (hit >= 0).sum() > len(sel) // 10
        assert_bits(hit_s, hit, f"{what}: ids")
        assert_bits(rec_s, rec, f"{what}: hit records")
    # exactly the maximum is accepted, one more is rejected with a clear error, and the engine keeps working
    n_max = rt.engine.TRACE_STATIC_MAX_RAYS
    rep = np.resize(np.arange(len(o)), n_max + 1)
    hit_s, rec_s = e.trace_static(o[rep[:n_max]], d[rep[:n_max]])
    hit, rec = e.trace(o[rep[:n_max]], d[rep[:n_max]])
    assert_bits(hit_s, hit, "maximum size: ids")
    assert_bits(rec_s, rec, "maximum size: hit records")
    with pytest.raises(rt.RtError, match="at most"):
        e.trace_static(o[rep], d[rep])
    hit_s, _ = e.trace_static(o[small], d[small])
    assert_bits(hit_s, e.trace(o[small], d[small])[0], "after the rejected call")
    assert e.trace_static(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))[0].size == 0
