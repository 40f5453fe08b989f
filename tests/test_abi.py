"""CPU: the C-ABI library loads and exports every symbol the headers in include/ declare (no compute calls)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exported():
    from cuda_optical_flow_2_amd import build, lib

    build.build()
    out = subprocess.run(["nm", "-D", "--defined-only", lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    return {line.split()[-1] for line in out.splitlines() if line.strip()}


def test_library_loads_and_reports_abi():
    from cuda_optical_flow_2_amd import lib

    L = lib.load()
    assert L.ofx_abi_version() == 5
    assert isinstance(L.ofx_last_error(), bytes)


def test_every_ofx_h_function_is_exported(exported):
    text = open(os.path.join(ROOT, "include", "ofx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(ofx_[a-z0-9_]+)\s*\(", text))
    assert len(declared) > 30
    missing = sorted(declared - exported)
    assert not missing, f"declared in include/ofx.h but not exported: {missing}"


def test_binding_table_matches_header():
    from cuda_optical_flow_2_amd import lib

    text = open(os.path.join(ROOT, "include", "ofx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(ofx_[a-z0-9_]+)\s*\(", text))
    assert declared == set(lib.EXPORTS), sorted(declared ^ set(lib.EXPORTS))


def test_reference_cpp_surface_is_exported(exported):
    """The mangled gpu::/utils:: symbols and mask tables main.cu links against."""
    from cuda_optical_flow_2_amd import compat

    want = set(compat.GPU_SYMBOLS.values()) | set(compat.UTILS_SYMBOLS.values()) | set(compat.MASK_SYMBOLS)
    missing = sorted(want - exported)
    assert not missing, missing
    assert "gpu_compat_last_status" in exported


def test_struct_layout_matches_header():
    """ctypes mirrors of ofx_geom / ofx_params have the sizes a C compiler gives the header's structs."""
    import ctypes
    import tempfile

    from cuda_optical_flow_2_amd import lib

    src = '#include <stdio.h>\n#include "ofx.h"\nint main(){printf("%zu %zu\\n", sizeof(ofx_geom), sizeof(ofx_params));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        a, b = map(int, subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split())
    assert ctypes.sizeof(lib.Geom) == a and ctypes.sizeof(lib.Params) == b


def test_product_does_not_import_oracle():
    """The product package must never reach into oracle/ (the oracle is the checker, not a fallback)."""
    pkg = os.path.join(ROOT, "cuda_optical_flow_2_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "ofx_oracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, f


def test_suggested_frames_per_launch():
    """Host logic of engine.suggest_stream_batch: bounded by OFX_MAX_LK_ITEMS and by the pipeline's working set."""
    from cuda_optical_flow_2_amd.engine import suggest_stream_batch
    from cuda_optical_flow_2_amd.parallel import ShardPlan

    assert suggest_stream_batch(3840, 2160, 5, None, borrow_frames=True) == 4   # the measured optima
    assert suggest_stream_batch(3840, 2160, 5, None, borrow_frames=False) == 2
    assert suggest_stream_batch(1920, 1080, 4, None) == 8
    assert suggest_stream_batch(7680, 4320, 6, None) == 2      # never below two frames per launch
    assert suggest_stream_batch(3840, 2160, 5, ShardPlan(3840, 2160, 5, 9, 3, 8), True) == 8
    for w, h, L in [(640, 480, 7), (7680, 4320, 6), (64, 64, 2)]:
        assert suggest_stream_batch(w, h, L, None) * L <= 40   # OFX_MAX_LK_ITEMS
