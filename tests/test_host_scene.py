"""Host side of the product (no GPU): the C ABI library loads and exports every symbol of include/rt_engine.h, the
asset decoders and the scene builder reproduce the reference's data bit for bit (golden vectors), error behaviour."""
import ctypes
import hashlib
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ASSETS, ROOT, assert_bits, build_table_scene, synthetic_tables


def header_functions():
    text = open(os.path.join(ROOT, "include", "rt_engine.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(rt):
    E = rt.engine
    names = header_functions()
    assert len(names) > 40
    L = ctypes.CDLL(E.LIB_PATH)
    for n in names:
        assert hasattr(L, n), f"{n} is declared in include/rt_engine.h but not exported by librt_engine.so"
    assert sorted(E.SIGNATURES) == names, "the Python binding covers exactly the declared entry points"


def test_library_contains_gfx950_code(rt):
    blob = open(rt.engine.LIB_PATH, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in blob, "the fat binary holds no gfx950 code object"


def test_asset_decoders_match_the_reference(rt, golden):
    E = rt.engine
    g = golden("assets")
    v = E.load_obj(os.path.join(ASSETS, "tree.obj"))
    assert_bits(v, g["tree_verts8"], "LoadOBJ output")
    tex, fc = E.load_png(os.path.join(ASSETS, "tree_texture.png"), 4)      # RGBA, Adam7 interlaced
    assert fc == int(g["tree_texture_file_channels"]) and list(tex.shape) == list(g["tree_texture_shape"])
    assert hashlib.sha256(tex.tobytes()).digest() == g["tree_texture_sha256"].tobytes()
    assert_bits(tex[::37, ::41], g["tree_texture_probe"], "texture probe")
    sky, fc = E.load_png(os.path.join(ASSETS, "skybox12.png"), 3)
    assert fc == 3 and hashlib.sha256(sky.tobytes()).digest() == g["skybox_sha256"].tobytes()
    assert_bits(sky[::17, ::19], g["skybox_probe"], "skybox probe")
    # channel conversion and flip
    rgba_sky, _ = E.load_png(os.path.join(ASSETS, "skybox12.png"), 4)
    assert (rgba_sky[..., 3] == 255).all() and np.array_equal(rgba_sky[..., :3], sky)
    unflipped, _ = E.load_png(os.path.join(ASSETS, "skybox12.png"), 3, flip=False)
    assert np.array_equal(unflipped[::-1], sky)


def test_default_scene_matches_the_reference(rt, golden, build_key):
    g = golden("scene_default")
    s = rt.default_scene(build_key)
    for a, n in zip(s.dump_top(), ("top_boxes", "top_left", "top_right", "top_leaf_ids")):
        assert_bits(a, g[n], n)
    for a, n in zip(s.dump_mesh(2), ("mesh_boxes", "mesh_left", "mesh_right", "mesh_tris")):
        assert_bits(a, g[n], n)
    assert_bits(s.dump_objects(19), g["objects"], "objects")
    assert_bits(s.dump_instance(2), g["instance"], "instance")
    info = s.flat_info()
    # 19 inner + 18 leaves at the top; 511 inner + 457 triangles in the mesh; duplicate leaves listed once
    assert info["triangles"] == 457 and info["instances"] == 1 and info["objects"] == 19 and info["planes"] == 1
    assert info["entries"] == 19 + 18 + 511 + 457 + 3 + 1  # + END of the top list, END of the mesh range, IDLE; + the entry box of the tree's instance
    other = rt.default_scene(build_key + 1)
    assert not np.array_equal(other.dump_top()[1], g["top_left"]) or not np.array_equal(other.dump_top()[3], g["top_leaf_ids"])


def test_camera_matches_the_reference(rt, golden):
    assert_bits(rt.camera_look((0, 3, -10), (0, 0, 1)), golden("scene_default")["camera"], "camera basis")
    path = golden("camera_path")
    cam = rt.camera_look((0, 3, -10), (0, 0, 1))
    for f in range(240):
        cam = rt.camera_input(cam, 3.0, 1.0, w=int(f < 120), d=int(f >= 120))
        assert_bits(cam, path["cams"][f], f"camera after scripted frame {f}")


def test_custom_and_synthetic_bvh_match_the_reference(rt, golden, assets):
    g = golden("scene_custom")
    s = build_table_scene(rt.Scene(), g["tex"], g["mat"], g["obj"], g["plane"], assets["verts"], assets["tex"], assets["sky"], int(g["build_key"]))
    assert s.flat_info()["instances"] == 2
    g = golden("scene_synthetic")
    obj, tex, mat = synthetic_tables(int(g["n"]))
    s = build_table_scene(rt.Scene(), tex, mat, obj, [[0.0, len(mat) - 1]], None, None, assets["sky"], int(g["build_key"]))
    top = s.dump_top()
    assert hashlib.sha256(top[0].tobytes()).digest() == g["top_boxes_sha256"].tobytes()
    assert_bits(top[1], g["top_left"], "left links")
    assert_bits(top[2], g["top_right"], "right links")
    assert_bits(top[3], g["top_leaf_ids"], "leaf order")


def test_tie_heavy_scene_same_tree_as_oracle(rt, orc, assets):
    """grid-aligned boxes: thousands of equal sort keys — std::sort in the builder and its restatement in the oracle
    must order them identically"""
    rng = np.random.default_rng(9)
    n = 5000
    obj = np.zeros((n, 10), np.float32)
    obj[:, 0] = 2
    obj[:, 1:4] = rng.integers(0, 6, (n, 3)) * 2.0
    obj[:, 4] = 0.5
    tex = np.array([[0, .5, .5, .5, 0, 0, 0]], np.float32)
    mat = np.array([[1, 0, 0, 0, 0]], np.float32)
    a = build_table_scene(rt.Scene(), tex, mat, obj, [[0.0, 0]], None, None, assets["sky"], 77)
    b = build_table_scene(orc.Scene(), tex, mat, obj, [[0.0, 0]], None, None, assets["sky"], 77)
    for x, y, nme in zip(a.dump_top(), b.dump_top(), ("boxes", "left", "right", "leaf order")):
        assert_bits(x, y, nme)


def test_error_behaviour_without_gpu(rt):
    E = rt.engine
    s = rt.Scene()
    with pytest.raises(rt.RtError, match="material"):
        s.sphere((0, 0, 0), 1.0, 3)
    with pytest.raises(rt.RtError, match="texture"):
        s.material(E.MAT_DIFFUSE, -1)
    with pytest.raises(rt.RtError, match="cannot open"):
        s.texture_file("/nonexistent/texture.png")
    with pytest.raises(rt.RtError, match="cannot open"):
        s.mesh_file("/nonexistent/mesh.obj", 1.0, s.material(E.MAT_DIFFUSE, s.texture_color((1, 1, 1))))
    with pytest.raises(rt.RtError, match="no bounded objects"):
        rt.Scene().build(1)
    t = s.texture_color((1, 0, 0))
    s.sphere((0, 0, 0), 1.0, s.material(E.MAT_DIFFUSE, t))
    with pytest.raises(rt.RtError, match="skybox"):
        s.build(1)
    s.skybox(np.zeros((2, 4, 3), np.uint8))
    s.build(1)
    with pytest.raises(rt.RtError, match="already built"):
        s.build(1)
    with pytest.raises(rt.RtError):
        E.load_png(os.path.join(ASSETS, "tree.obj"), 3)          # not a PNG


def test_damaged_png_files_are_errors(rt, tmp_path):
    """a file cut short, a header announcing more pixels than the data can hold, and a flipped data byte are all turned
    away with an error (nothing of the announced size is allocated first, no exception leaves the library)"""
    import struct
    import zlib
    E = rt.engine
    good = open(os.path.join(ASSETS, "tree_texture.png"), "rb").read()
    cases = {"cut.png": good[:len(good) // 2], "cut_header.png": good[:20], "flipped.png": good[:4000] + bytes([good[4000] ^ 0x55]) + good[4001:]}
    # the same file with a header announcing 30000 x 30000 pixels
    ihdr = struct.pack(">II", 30000, 30000) + good[24:29]
    cases["huge.png"] = good[:12] + b"IHDR" + ihdr + struct.pack(">I", zlib.crc32(b"IHDR" + ihdr)) + good[33:]
    for name, data in cases.items():
        p = tmp_path / name
        p.write_bytes(data)
        with pytest.raises(rt.RtError):
            E.load_png(str(p), 4)


def test_engine_refuses_to_run_without_a_gpu(rt):
    """no CPU fallback: creating an engine without a HIP device is an error"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rt.RtError):
        rt.Engine(rt.default_config(width=16, height=16))


def test_front_end_adapter_compiles_against_the_front_end_headers(tmp_path):
    """The FrameManager / World / Camera adapter must compile with the front end's own Constants.h and GLM (both stay in
    the front end's tree, so this runs only where that tree is present)."""
    import os
    import subprocess
    from conftest import ROOT
    front, glm = "/root/reference/Raytracer/src", "/root/reference/glm/glm"
    if not (os.path.exists(front + "/Constants.h") and os.path.exists(glm + "/glm.hpp")):
        pytest.skip("front-end sources are not on this machine")
    comp = os.path.join(ROOT, "raytracer_amd", "csrc", "compat")
    for src in ("compat.cpp", "headless_main.cpp"):
        subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I" + comp, "-I" + front, "-I" + glm,
                               os.path.join(comp, src)])


def test_fog_scene_topology_matches_the_reference(golden, assets):
    import raytracer_amd as rt
    g = golden("scene_fog")
    s = build_table_scene(rt.Scene(), g["tex"], g["mat"], g["obj"], g["plane"], None, None, assets["sky"], int(g["build_key"]))
    top = s.dump_top()
    assert_bits(top[1], g["top_left"], "fog scene BVH left links vs the reference")
    assert_bits(top[2], g["top_right"], "fog scene BVH right links vs the reference")
    assert_bits(top[3], g["top_leaf_ids"], "fog scene BVH leaf order vs the reference")
    with pytest.raises(rt.RtError):
        bad = rt.Scene()
        bad.fog((0, 0, 0), 1.0, 1.0, bad.material(1, bad.texture_color((1, 1, 1)), (0, 0, 0)))   # not an isotropic material


def test_alt_scene_is_the_first_six_rows_of_the_fog_fixture(golden, assets):
    """rt_scene_add_alt_objects builds World.cpp:34-40 at time 0; the fog fixture's first six rows are that scene as given
    to the reference, so the object records must agree"""
    import raytracer_amd as rt
    g = golden("scene_fog")
    a = rt.alt_scene(123)
    b = build_table_scene(rt.Scene(), g["tex"], g["mat"], g["obj"][:6], g["plane"], None, None, assets["sky"], 123)
    assert_bits(a.dump_objects(7)[:6], b.dump_objects(7)[:6], "alternative scene objects")     # the fixture's plane has other colours
    assert a.dump_objects(7)[6][14:17].tolist() == [np.float32(0.2), np.float32(0.6), np.float32(0.3)]
    for x, y, n in zip(a.dump_top(), b.dump_top(), ("boxes", "left", "right", "leaf")):
        assert_bits(x, y, "alternative scene BVH " + n)
    moved = rt.alt_scene(123, time=7.0)
    assert moved.dump_objects(7)[4][1] == np.float32(3 * np.sin(np.float64(np.float32(7.0) * np.float32(0.1))))


def test_c3_tree_built_by_the_product(rt, golden, assets):
    """the host builder on the 100 000-primitive scene: the reference's tree (digests of boxes, links, leaf order)"""
    import time
    g = golden("scene_c3")
    obj, tex, mat = synthetic_tables(int(g["n"]))
    t0 = time.perf_counter()
    s = build_table_scene(rt.Scene(), tex, mat, obj, [[0.0, len(mat) - 1]], None, None, assets["sky"], int(g["build_key"]))
    dt = time.perf_counter() - t0
    for name, got in zip(("boxes", "left", "right", "leaf_ids"), s.dump_top()):
        assert_bits(got[::1000], g[f"top_{name}_rows"], f"C3 tree {name}, every 1000th row")
        assert hashlib.sha256(np.ascontiguousarray(got).tobytes()).digest() == g[f"top_{name}_sha256"].tobytes(), f"C3 tree {name}"
    print(f"C3 scene through the C ABI: {dt:.2f} s including {int(g['n'])} object and material calls from Python")


@pytest.mark.parametrize("layout", [1, 2])
@pytest.mark.parametrize("key", [0x12345678, 0x5EED0001, 7])
def test_regrouped_layout_keeps_the_lowest_nodes_and_their_leaves(rt, key, layout):
    """The regrouped layout (the levels of a mesh's tree above its lowest nodes rebuilt): every leaf and every lowest node
    of the reference's layout is there with the same bits, leaves keep their ranks, every box of the new upper levels
    contains the boxes below it (which is what makes it visit every lowest node the reference visits), and the top-level
    tree — whose boxes over negative radii are not even ordered — is left alone.  layout 2: the same with wide nodes (a
    box of more than half its parent's surface left out, its children listed in its place): fewer entries, nodes of up to
    six children, and by the surface measure fewer tests again."""
    ENT_INNER, ENT_TRI, ENT_INSTANCE = 0, 3, 8
    s = rt.default_scene(key)
    info = s.regroup_info()
    assert info["has_regrouped_layout"] == 1 and info["trees"] == 1 and info["lowest_nodes"] == 256
    assert info["mesh_tests_regrouped"] < 0.8 * info["mesh_tests_reference"], info
    assert info["top_tests_regrouped"] == info["top_tests_reference"]
    ref, ref_peer, ref_rank = s.dump_entries(False)
    new, new_peer, new_rank = s.dump_entries(layout)
    if layout == 1:
        assert len(ref) == len(new)                                    # a binary tree over the same lowest nodes
    else:
        wide = s.wide_info()
        assert wide["has_wide_layout"] == 1 and wide["entries"] == len(new) < len(ref) - 100
        assert 3 <= wide["most_children"] <= 8 and wide["mesh_tests_wide"] < 0.95 * info["mesh_tests_regrouped"]
    kind = lambda e: e[:, 7].view(np.int32)
    link = lambda e: e[:, 3].view(np.int32)
    leaf_r = np.nonzero((kind(ref) >= 1) & (kind(ref) <= 4))[0]
    leaf_n = np.nonzero((kind(new) >= 1) & (kind(new) <= 4))[0]
    assert len(leaf_r) == len(leaf_n) and (kind(ref) == ENT_TRI).sum() == 457
    if layout == 2:
        # (a leaf of the reference's layout names its entry in the regrouped layout without wide nodes; the way from there
        # to this layout is this layout's own peers, read backwards)
        ref_peer = np.full(len(ref), -1, np.int32)
        ref_peer[new_peer[leaf_n]] = leaf_n
    # leaves: same data, same rank, peers point at each other; ranks number the leaves in the reference's order
    assert sorted(ref_peer[leaf_r].tolist()) == leaf_n.tolist()
    assert (new_peer[ref_peer[leaf_r]] == leaf_r).all()
    assert_bits(new[ref_peer[leaf_r]], ref[leaf_r], "leaf entries of the two layouts")
    assert (new_rank[ref_peer[leaf_r]] == ref_rank[leaf_r]).all()
    assert sorted(ref_rank[leaf_r].tolist()) == list(range(len(leaf_r)))
    # lowest nodes (inner entries followed by leaves only): the same boxes, each still followed by its own leaves
    def lowest(e):
        out = {}
        k, ln = kind(e), link(e)
        for i in np.nonzero(k == ENT_INNER)[0]:
            kids = list(range(i + 1, ln[i]))
            if kids and all(1 <= k[c] <= 4 for c in kids):
                out[i] = kids
        return out
    low_r, low_n = lowest(ref), lowest(new)
    by_leaves = {tuple(ref_peer[c] for c in kids): i for i, kids in low_r.items()}
    assert len(low_n) == len(low_r)
    for j, kids in low_n.items():
        i = by_leaves[tuple(kids)]
        assert ref[i, [0, 1, 2, 4, 5, 6]].tobytes() == new[j, [0, 1, 2, 4, 5, 6]].tobytes()
    # containment, in the instance's range of the regrouped layout
    k, ln = kind(new), link(new)
    inner = np.nonzero(k == ENT_INNER)[0]
    top_end = int(np.nonzero(kind(ref) == 5)[0][0])
    assert_bits(new[:top_end + 1], ref[:top_end + 1], "top-level list")
    checked = 0
    for i in inner[inner > top_end]:
        c = i + 1
        while c < ln[i]:
            if k[c] == ENT_INNER:
                assert (new[i, 0:3] <= new[c, 0:3]).all() and (new[i, 4:7] >= new[c, 4:7]).all(), (i, c)
                checked += 1
                c = ln[c]
            else:
                c += 1
    assert checked >= (500 if layout == 1 else 350)
