"""Build libecwam_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
from __future__ import annotations

import hashlib
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libecwam_hip.so")
SOURCES = ["capi.hip", "propag.hip", "implsch4.hip", "implsch4a.hip", "implsch4x.hip", "implsch4r.hip", "implsch4rd.hip", "outbs.hip"]
# objects that are a second compilation of another source: object name -> (source, extra flags; a later -O overrides the earlier one).
# implsch4rd = the double precision RARE builds of k_implsch4 as the two-kernel split (V4R_DP = 2) at -O3: as ONE function they fault on the device
# at -O3 (implsch4r.hip; round 5 shipped that function at -O2) -- the split passes at every optimisation level and is bit-identical
DERIVED = {"implsch4r.hip": ("implsch4r.hip", ["-DV4R_PREC=1"]), "implsch4rd.hip": ("implsch4r.hip", ["-DV4R_PREC=2", "-DV4R_DP=2"])}
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
# IMPLSCH is VALU-issue bound: single-precision divide/sqrt by the hardware reciprocal / square root plus one
# refinement (<= 2.5 ulp) instead of the correctly rounded sequences; double precision is unaffected.
FAST_DIV = ["-fno-hip-fp32-correctly-rounded-divide-sqrt"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-variable", "-Wno-unused-but-set-variable"]
IMPLSCH_SOURCES = ("implsch4.hip", "implsch4a.hip", "implsch4x.hip", "implsch4r.hip", "implsch4rd.hip")
# Build variants of the IMPLSCH translation units (the single-precision error attribution):
#   ""         the product build: hardware reciprocal / square root / exp2 / log2 in single precision, FMA contraction on
#   "exactdiv" every `/` and SQRT the source spells out is correctly rounded (the scalar chains per sea point: TAUT_Z0, STRESSO, FKMEAN,
#              FRCUTINDEX ...); the per-bin sites of the row loops, which call f_div / f_rcp / f_rsq / f_sqrt, keep the hardware approximations
#   "strict1"  ECWAM_HIP_STRICT=1: those per-bin divisions and square roots correctly rounded as well
#   "site4/8/32"  exactdiv + ONE group of per-bin sites exact: 4 the reciprocal of COS(TH - wind stress direction) in SINPUT_ARD's ZLOG,
#              8 the row-uniform chain of the sheltering (1 / |TAU|, SQRT, 1 / (U*/c + ZALP)), 32 DFIM / SQRT(WAVNUM) of FKMEAN.  (The
#              experiment's fourth group, 16 = the divisions of the implicit update, was the one that mattered and became the product's
#              refined quotient dev.h::f_div_r: profiles/r03_sp_error_attribution.txt.)
#   "strict2"  ECWAM_HIP_STRICT=2: EXP / LOG with an exact argument reduction (error of the hardware exp2 / log2 only)
#   "strict3"  both
#   "strict7"  both + floating-point contraction off (no FMA the source does not spell out)
# A variant is written to libecwam_hip_<variant>.so next to the product library; ECWAM_HIP_LIB=<path> makes lib.load() use it.
VARIANTS = {"": FAST_DIV, "exactdiv": ["-DECWAM_HIP_STRICT=0"], "strict1": ["-DECWAM_HIP_STRICT=1"], "strict2": FAST_DIV + ["-DECWAM_HIP_STRICT=2"],
            "site4": ["-DECWAM_HIP_STRICT=4"], "site8": ["-DECWAM_HIP_STRICT=8"], "site32": ["-DECWAM_HIP_STRICT=32"],
            "strict3": ["-DECWAM_HIP_STRICT=3"], "strict7": ["-DECWAM_HIP_STRICT=3", "-ffp-contract=off"],
            "noieee": FAST_DIV + ["-mno-amdgpu-ieee", "-fno-honor-nans"],
            # the double precision RARE builds of k_implsch4 (profiles/r05_rare_dp_rootcause.txt, r06_rare_dp_note.txt): the object implsch4rd
            # (product: the two-kernel split at -O3, V4R_DP = 2) as the one kernel at -O3 (faults), at -O3 with index assertions, at -O2
            "rdp": FAST_DIV + ["-O3", "-DV4R_DP=1"], "rdpchk": FAST_DIV + ["-O3", "-DV4R_DP=1", "-DV4_CHECK=1"], "rdpO2": FAST_DIV + ["-O2", "-DV4R_DP=1"],
            # every build of k_implsch4 as the two-kernel split (PART 1: through the second SINFLX call | PART 2: sweep, fluxes, tail, stores)
            "split": FAST_DIV + ["-DV4_SPLIT_ALL=1", "-DV4R_DP=2"],
            # A/B partners of the mechanisms of round 5 that are in the product (profiles/r05_scalar_prefetch.txt): the per-interaction / per-row
            # coefficient records fetched ahead (V4_RECPF, single precision) off; the double precision sweep's record as the compiler places its
            # loads (V4_RECV = 0); the all-reduces of a SINPUT row one after the other instead of in one batch (V4_REDN = 0)
            "norecpf": FAST_DIV + ["-DV4_RECPF=0"], "norecs": FAST_DIV + ["-DV4_RECV=0"], "noredn": FAST_DIV + ["-DV4_REDN=0"],
            # round 6: the library with the go / no-go probe of the one-kernel step (implsch4a.hip, flags bit 1 of ecwam_hip_propags2_implsch)
            "advprobe": FAST_DIV + ["-DV4_ADV_PROBE=1"],
            # the on-the-fly CTU weights (k_propags2_otf and the advecting load of k_implsch4) in ctuw.F90's order of operations, contraction off:
            # bit-identical to the stored-weight scheme (csrc/ctu.h; the product hoists the factors and fuses the multiply-adds)
            "ctustrict": FAST_DIV,
            # the advecting load with 2 / 4 steps of gathers in flight instead of 3 (V4_ADV_DEPTH), and at a raised wave priority (s_setprio):
            # all within the noise of the product (profiles/r06_fused_step_experiments.txt)
            # the one-kernel builds at -O1 (with every double precision direction count enabled in implsch4a.hip this was the proof that their -O3
            # failures are code generation: profiles/r06_fused_step_experiments.txt)
            "advO1": FAST_DIV + ["-O1"],
            "advd2": FAST_DIV + ["-DV4_ADV_DEPTH=2"], "advd4": FAST_DIV + ["-DV4_ADV_DEPTH=4"], "advprio": FAST_DIV + ["-DV4_ADV_PRIO=2"]}
# flags a variant adds to EVERY source it rebuilds (not only the IMPLSCH units)
VARIANT_ANY = {"ctustrict": ["-DECWAM_HIP_CTU_STRICT=1"]}
# variants that rebuild only some of the translation units (the other objects are the product's)
VARIANT_SOURCES = {"ctustrict": ("propag.hip", "implsch4a.hip"), "advprobe": ("implsch4a.hip",), "advO1": ("implsch4a.hip",), "advd2": ("implsch4a.hip",), "advd4": ("implsch4a.hip",),
                   "advprio": ("implsch4a.hip",), "rdp": ("implsch4rd.hip",), "rdpchk": ("implsch4rd.hip",), "rdpO2": ("implsch4rd.hip",),
                   "split": ("implsch4.hip", "implsch4x.hip", "implsch4r.hip", "implsch4rd.hip"), "norecpf": ("implsch4.hip",), "norecs": ("implsch4.hip",),
                   "noredn": ("implsch4.hip",)}

INCLUDE = os.path.join(HERE, "..", "include", "ecwam_hip.h")


def lib_path(variant: str = "") -> str:
    return LIB if not variant else os.path.join(LIBDIR, f"libecwam_hip_{variant}.so")


def _closure(path: str, seen: dict) -> None:
    """The file and every header it includes with quotes, recursively (system headers come with the toolchain)."""
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return
    with open(path, "rb") as fh:
        data = fh.read()
    seen[path] = data
    for m in re.finditer(rb'^[ \t]*#[ \t]*include[ \t]*"([^"]+)"', data, re.M):
        _closure(os.path.join(os.path.dirname(path), m.group(1).decode()), seen)


def _stamp(src: str, flags: list) -> str:
    """Content hash of everything an object depends on: the command line, the source and the closure of its quoted includes.
    (mtime comparisons reuse stale objects that travelled with a snapshot, and miss a header that is not in a hand-kept list.)"""
    seen: dict = {}
    _closure(os.path.join(CSRC, src), seen)
    h = hashlib.sha256(" ".join([HIPCC, *flags]).encode())
    for k in sorted(seen):
        h.update(os.path.relpath(k, HERE).encode())
        h.update(seen[k])
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False, variant: str = "") -> str:
    """Compile the objects whose command line, source or included headers changed (content hashes in <obj>.stamp), then link."""
    if variant not in VARIANTS:
        raise ValueError(f"unknown build variant {variant!r}: {sorted(VARIANTS)}")
    os.makedirs(LIBDIR, exist_ok=True)
    lib = lib_path(variant)
    objs, procs, stamps = [], [], {}
    for src in SOURCES:
        special = bool(variant) and src in VARIANT_SOURCES.get(variant, IMPLSCH_SOURCES)
        obj = os.path.join(LIBDIR, src.replace(".hip", f".{variant}.o" if special else ".o"))
        real_src, extra = DERIVED.get(src, (src, []))
        flags = FLAGS + extra + ((VARIANTS[variant] if special else FAST_DIV) if src in IMPLSCH_SOURCES else []) + (VARIANT_ANY.get(variant, []) if special else [])
        objs.append(obj)
        st = _stamp(real_src, flags)
        old = ""
        if os.path.exists(obj) and os.path.exists(obj + ".stamp"):
            with open(obj + ".stamp") as fh:
                old = fh.read().strip()
        if not force and old == st:
            continue
        stamps[obj] = st
        # -cuid: clang derives a compilation-unit id from the source file's PATH by default, which goes into the code object's symbol
        # names -- the library's bytes (and the sha256 bench.py ties the counter summaries to) would depend on where the tree lives
        cmd = [HIPCC, *flags, f"-cuid=ecwam_{os.path.basename(obj).replace('.', '_')}", "-c", os.path.join(CSRC, real_src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, obj, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{out}")
        with open(obj + ".stamp", "w") as fh:
            fh.write(stamps[obj])
        if verbose and out.strip():
            print(out)
    link_stamp = hashlib.sha256("".join(open(o + ".stamp").read() for o in objs).encode()).hexdigest()
    if not procs and os.path.exists(lib) and os.path.exists(lib + ".stamp") and open(lib + ".stamp").read().strip() == link_stamp:
        return lib
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    with open(lib + ".stamp", "w") as fh:
        fh.write(link_stamp)
    return lib


FLANG = shutil.which("amdflang") or "/opt/rocm/lib/llvm/bin/flang"
FSRC = os.path.join(HERE, "fortran")
# the Fortran host layer: C interfaces + device state + transfer workers, the generated host types (tools/gen_yowdrvtype.py), the set-up
# layer, the driver; then the harness programs (one executable each)
FFILES = ["ecwam_hip_capi.F90", "yowdrvtype_hip.F90", "ecwam_hip_mod.F90", "ecwam_hip_restart.F90", "wamintgr_hip.F90", "harness_case.F90"]
FPROGS = {"smoke_wamintgr_hip": "smoke_wamintgr_hip.F90", "seam_sequence": "seam_sequence.F90"}


def fortran_exe(prec: str, prog: str = "smoke_wamintgr_hip") -> str:
    return os.path.join(LIBDIR, f"{prog}_{prec}")


def build_fortran(force: bool = False) -> list:
    """Fortran host layer (iso_c_binding module + host types + WAMINTGR_HIP) and the harness programs, sp and dp, linked against
    libecwam_hip.so with amdflang."""
    build(force=False)
    out = []
    for prec in ("sp", "dp"):
        exes = [fortran_exe(prec, p) for p in FPROGS]
        out += exes
        srcs = [os.path.join(FSRC, f) for f in FFILES + list(FPROGS.values())]
        if not force and all(os.path.exists(e) and all(os.path.getmtime(s) < os.path.getmtime(e) for s in srcs + [LIB]) for e in exes):
            continue
        moddir = os.path.join(LIBDIR, f"fmod_{prec}")
        os.makedirs(moddir, exist_ok=True)
        defs = ["-DECWAM_HIP_SINGLE"] if prec == "sp" else []

        def compile_(s):
            o = os.path.join(moddir, os.path.basename(s).replace(".F90", ".o"))
            r = subprocess.run([FLANG, "-cpp", "-O2", "-fPIC", *defs, "-module-dir", moddir, "-I", moddir, "-c", s, "-o", o],
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"flang failed for {s}:\n{r.stdout}")
            return o

        objs = [compile_(os.path.join(FSRC, f)) for f in FFILES]
        for prog, f in FPROGS.items():
            po = compile_(os.path.join(FSRC, f))
            r = subprocess.run([FLANG, "-o", fortran_exe(prec, prog), po, *objs, "-L", LIBDIR, "-lecwam_hip", f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,$ORIGIN"],
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"fortran link failed ({prog}):\n{r.stdout}")
    return out


if __name__ == "__main__":
    vs = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--variant=")]
    for v in vs or [""]:
        print(build(force="--force" in sys.argv, verbose=True, variant=v))
    if not vs:
        print(build_fortran(force="--force" in sys.argv))
