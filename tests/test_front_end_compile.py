"""The reference's front end, unchanged, against the adapter headers (raytracer_amd/csrc/compat).

App.h says `#include "FrameManager.h"` and App.cpp `#include "World.h"` (App.h:2, App.cpp:2): a quoted include looks in
the including file's own directory first, so the adapter's headers are found only when the reference's engine headers
(FrameManager.h, World.h, DataUtility.h, Object.h) are NOT beside App.h — putting compat/ in front on the include path
is not enough.  The test reproduces the layout a maintainer is told to use in a scratch directory: the four front-end
files alone, the adapter directory on the include path; it compiles them (syntax and semantics, no code generation —
there is no SDL2 library for Linux here, its headers ship with the reference) and asks the preprocessor which headers it
used.  Nothing is copied into the repository."""
import os
import shutil
import subprocess

import pytest

REF = "/root/reference"
SRC = os.path.join(REF, "Raytracer", "src")
GLM = os.path.join(REF, "glm", "glm")
SDL = os.path.join(REF, "SDL2", "include")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMPAT = os.path.join(ROOT, "raytracer_amd", "csrc", "compat")
FRONT_END = ("App.h", "App.cpp", "Main.cpp", "Constants.h")
ENGINE_HEADERS = ("FrameManager.h", "World.h", "DataUtility.h")

pytestmark = pytest.mark.skipif(not (os.path.exists(os.path.join(SRC, "App.cpp")) and os.path.isdir(SDL) and shutil.which("g++")),
                                reason="needs the reference's front-end sources, its SDL2 headers and g++")


def headers_used(make_rule, cwd):
    return {os.path.realpath(os.path.join(cwd, p)) for p in make_rule.replace("\\\n", " ").split()[1:]}


def gxx(front, *args):
    cmd = ["g++", "-std=c++17", "-w", "-I", COMPAT, "-I", ".", "-I", os.path.join(ROOT, "include"), "-I", GLM, "-I", SDL, *args]
    return subprocess.run(cmd, cwd=front, capture_output=True, text=True)


@pytest.fixture(scope="module")
def front(tmp_path_factory):
    d = tmp_path_factory.mktemp("front_end")
    for name in FRONT_END:
        shutil.copy(os.path.join(SRC, name), d / name)
    return str(d)


@pytest.mark.parametrize("unit", ["App.cpp", "Main.cpp"])
def test_unchanged_front_end_compiles_against_the_adapter(front, unit):
    r = gxx(front, "-fsyntax-only", unit)
    assert r.returncode == 0, r.stderr[-2000:]
    deps = gxx(front, "-M", unit)
    assert deps.returncode == 0, deps.stderr[-2000:]
    used = headers_used(deps.stdout, front)
    for h in ENGINE_HEADERS:
        assert os.path.realpath(os.path.join(COMPAT, h)) in used, f"{unit} did not include the adapter's {h}"
    assert os.path.realpath(os.path.join(ROOT, "include", "rt_engine.h")) in used
    inside_ref_src = [p for p in used if p.startswith(os.path.realpath(SRC) + os.sep)]
    assert not inside_ref_src, f"{unit} picked up the reference's own headers: {inside_ref_src}"


def test_beside_the_reference_headers_the_include_path_alone_is_not_enough():
    """the layout a maintainer is warned about: compiled where it lies, App.cpp finds the reference's FrameManager.h next
    to App.h whatever the include path says"""
    deps = gxx(SRC, "-M", "App.cpp")
    assert deps.returncode == 0, deps.stderr[-2000:]
    assert os.path.realpath(os.path.join(SRC, "FrameManager.h")) in headers_used(deps.stdout, SRC)
