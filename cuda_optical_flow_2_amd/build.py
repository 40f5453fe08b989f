"""Build libofx_hip.so (hand-written HIP for gfx950 + the C ABI of include/ofx.h) in-tree with hipcc.

No torch involvement: the library depends only on the HIP runtime, so the same file serves ctypes (Python),
the C++ compat surface (include/OptFlowGpu.cuh) and any other FFI.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(ROOT, "include")
OUT = os.path.join(PKG, os.environ.get("OFX_BUILD_OUT", "libofx_hip.so"))  # experiments: alternative builds side by side
OBJ = os.path.join(PKG, "csrc", "_obj" + os.environ.get("OFX_BUILD_TAG", ""))
ARCH = "gfx950"

SOURCES = ["lk_inst_stream_f8.hip", "lk_inst_stream_fast8.hip", "lk_inst_stream_fw.hip", "lk_inst_stream_fastw.hip", "lk_inst_stream_fwr.hip", "lk_inst_stream_fastwr.hip", "lk_inst_iter_f4.hip", "lk_inst_iter_fast4.hip", "lk_inst_stream_f.hip", "lk_inst_stream_fast.hip", "lk_inst_stream_c.hip", "lk_inst_levels_f.hip", "lk_inst_levels_fast.hip",
           "lk_inst_levels_c.hip", "lk_inst_iter_f2.hip", "lk_inst_iter_fast2.hip", "lk_inst_iter_f3.hip", "lk_inst_iter_fast3.hip", "lk_inst_iter_f1.hip", "lk_inst_iter_fast1.hip", "lk_level.hip", "corner.hip", "pyr_corner.hip", "pyramid.hip", "primitives.hip", "srm_march.hip", "ofx_core.cpp", "session.cpp",
           "compat_gpu.cpp", "compat_cpu.cpp", "compat_stage.cpp"]
# -ffp-contract=off: parity with the reference's x86-64 CPU build, which never fuses a*b+c (parity)
# -fno-slp-vectorize: hipcc otherwise packs scalar fp32 adds/fmas into v_pk_* pairs, which costs register moves and
# buys nothing on gfx950 (packed fp32 issues at half the rate of scalar fp32; tools/ubench/valu_rates.hip)
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", f"--offload-arch={ARCH}", "-I" + INCLUDE, "-I" + CSRC,
         "-Wall", "-Wno-unused-function"] + os.environ.get("OFX_BUILD_DEFS", "").split()


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP library cannot be built (there is no CPU fallback)")
    return exe


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE)]
    return hs


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    hdrs = _headers()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s + ".o")
        if force or _stale(obj, [src] + hdrs):
            lang = ["-x", "hip"] if s.endswith(".cpp") else []
            jobs.append((s, [cc] + FLAGS + lang + ["-c", src, "-o", obj]))

    def run(job):
        name, cmd = job
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {name}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return name

    if jobs:
        with ThreadPoolExecutor(max_workers=min(os.cpu_count() or 4, 8, len(jobs))) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s + ".o") for s in srcs]
    if force or jobs or _stale(OUT, objs):
        cmd = [cc, "-shared", "-fPIC", "-pthread", "-Wl,--no-undefined", f"--offload-arch={ARCH}", "-o", OUT] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
