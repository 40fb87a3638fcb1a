"""Build libecwam_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libecwam_hip.so")
SOURCES = ["capi.hip", "propag.hip", "implsch.hip", "implsch4.hip", "outbs.hip"]
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
# IMPLSCH is VALU-issue bound: single-precision divide/sqrt by the hardware reciprocal / square root plus one
# refinement (<= 2.5 ulp) instead of the correctly rounded sequences; double precision is unaffected.
EXTRA = {"implsch.hip": ["-fno-hip-fp32-correctly-rounded-divide-sqrt"], "implsch4.hip": ["-fno-hip-fp32-correctly-rounded-divide-sqrt"]}
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-variable", "-Wno-unused-but-set-variable"]


INCLUDE = os.path.join(HERE, "..", "include", "ecwam_hip.h")
DEPS = {"capi.hip": ["dev.h"], "propag.hip": ["dev.h"], "implsch.hip": ["dev.h", "implsch_common.h", "implsch_v2.h"], "implsch4.hip": ["dev.h", "implsch_common.h", "implsch_v2.h", "implsch_v4.h"], "outbs.hip": ["dev.h"]}


def _obj_stale(src: str, obj: str) -> bool:
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    deps = [os.path.join(CSRC, src), INCLUDE, os.path.abspath(__file__)] + [os.path.join(CSRC, d) for d in DEPS.get(src, [])]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the objects whose source (or a header it includes) changed, then link."""
    os.makedirs(LIBDIR, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        objs.append(obj)
        if not force and not _obj_stale(src, obj):
            continue
        cmd = [HIPCC, *FLAGS, *EXTRA.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    if not procs and os.path.exists(LIB) and all(os.path.getmtime(o) <= os.path.getmtime(LIB) for o in objs):
        return LIB
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    return LIB


FLANG = shutil.which("amdflang") or "/opt/rocm/lib/llvm/bin/flang"
FSRC = os.path.join(HERE, "fortran")
FFILES = ["ecwam_hip_mod.F90", "wamintgr_hip.F90", "smoke_wamintgr_hip.F90"]


def fortran_exe(prec: str) -> str:
    return os.path.join(LIBDIR, f"smoke_wamintgr_hip_{prec}")


def build_fortran(force: bool = False) -> list:
    """Fortran host layer (iso_c_binding module + WAMINTGR_HIP + harness program), sp and dp, linked against
    libecwam_hip.so with amdflang."""
    build(force=False)
    out = []
    for prec in ("sp", "dp"):
        exe = fortran_exe(prec)
        out.append(exe)
        srcs = [os.path.join(FSRC, f) for f in FFILES]
        if not force and os.path.exists(exe) and all(os.path.getmtime(s) < os.path.getmtime(exe) for s in srcs + [LIB]):
            continue
        moddir = os.path.join(LIBDIR, f"fmod_{prec}")
        os.makedirs(moddir, exist_ok=True)
        defs = ["-DECWAM_HIP_SINGLE"] if prec == "sp" else []
        objs = []
        for s in srcs:
            o = os.path.join(moddir, os.path.basename(s).replace(".F90", ".o"))
            r = subprocess.run([FLANG, "-cpp", "-O2", "-fPIC", *defs, "-module-dir", moddir, "-I", moddir, "-c", s, "-o", o],
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"flang failed for {s}:\n{r.stdout}")
            objs.append(o)
        r = subprocess.run([FLANG, "-o", exe, *objs, "-L", LIBDIR, "-lecwam_hip", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,$ORIGIN"],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"fortran link failed:\n{r.stdout}")
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_fortran(force="--force" in sys.argv))
