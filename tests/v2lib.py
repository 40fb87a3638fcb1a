"""TEST INFRASTRUCTURE: the second IMPLSCH implementation of the GPU tests -- k_implsch2 (tests/csrc/implsch_v2.h), the one-point-per-wavefront
kernel that was the product's kernel of rounds 1 - 4 for what k_implsch4 did not cover.  Built here into tests/csrc/libecwam_v2.so (hipcc,
gfx950; the product's csrc/ on the include path for dev.h and the shared point-wise routines) and launched on the device tables of a product
context (ecwam_hip_device_tables).  install() gives ecwam_amd.api.HipContext the two methods the tests use to choose between the
implementations; nothing under ecwam_amd/ knows this file."""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
PRODUCT_CSRC = os.path.join(ROOT, "ecwam_amd", "csrc")
LIB = os.path.join(CSRC, "libecwam_v2.so")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-fno-hip-fp32-correctly-rounded-divide-sqrt", "-Wno-unused-variable",
         "-Wno-unused-but-set-variable", "-I", PRODUCT_CSRC, "-I", CSRC]
_lib = None


def _stamp() -> str:
    h = hashlib.sha256(" ".join([HIPCC, *FLAGS]).encode())
    for d, names in ((CSRC, ("implsch_v2.hip", "implsch_v2.h", "implsch_wave_v2.h")), (PRODUCT_CSRC, ("dev.h", "implsch_common.h", "implsch_point.h")),
                     (os.path.join(ROOT, "include"), ("ecwam_hip.h",))):
        for n in names:
            with open(os.path.join(d, n), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()


def build(force: bool = False) -> str:
    st = _stamp()
    if not force and os.path.exists(LIB) and os.path.exists(LIB + ".stamp") and open(LIB + ".stamp").read().strip() == st:
        return LIB
    r = subprocess.run([HIPCC, *FLAGS, "-o", LIB, os.path.join(CSRC, "implsch_v2.hip")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for the tests' k_implsch2 library:\n{r.stdout}")
    with open(LIB + ".stamp", "w") as fh:
        fh.write(st)
    return LIB


def load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        vp, ci = C.c_void_p, C.c_int
        _lib.ecwam_v2_implsch.argtypes = [vp, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp, ci, ci, ci, vp]
        _lib.ecwam_v2_implsch.restype = ci
    return _lib


def variant_bits(cfg) -> int:
    """The build of k_implsch2 a configuration selects (what capi.hip packed for launch_implsch through round 4)."""
    rare = (cfg.lciwa1 or cfg.lciwa2 or cfg.lciwa3 or cfg.lciscal or cfg.lwnemocou or cfg.lwnemocouwrs or cfg.lwnemocoustrn or cfg.isnonlin
            or cfg.iphys == 0 or cfg.icode != 3 or not cfg.lwvflx_snl)
    return (16 if cfg.llnormagam else 0) | (32 if (cfg.llgcbz0 or rare) else 0)


def install():
    """ctx.set_implsch_generation(gen) and a ctx.implsch / ctx.implsch_generation_used pair that honour it: gen 2 = k_implsch2 of this
    library on the context's device tables, 0 / 4 = the product's own call."""
    import torch

    from ecwam_amd import api

    if getattr(api.HipContext, "_v2_installed", False):
        return
    product_implsch = api.HipContext.implsch
    product_used = api.HipContext.implsch_generation_used

    def set_implsch_generation(self, gen: int) -> None:
        if gen not in (0, 2, 4):
            raise ValueError("generation 0 (the product), 2 (the tests' k_implsch2) or 4")
        self._test_gen = int(gen)

    def implsch(self, kijs, kijl, fl1, wvprpt, ff, intf, mij, xllws, dbg=None, wam2nemo=None):
        if getattr(self, "_test_gen", 0) != 2:
            self._test_last = 0
            return product_implsch(self, kijs, kijl, fl1, wvprpt, ff, intf, mij, xllws, None, wam2nemo)
        cfg = self.t.cfg
        dp = 1 if fl1.dtype == torch.float64 else 0
        for t in (fl1, wvprpt, ff, intf, mij, xllws):
            assert t.is_cuda and t.is_contiguous()
        pw = wam2nemo.data_ptr() if (wam2nemo is not None and cfg.lwnemocou) else None
        pd = dbg.data_ptr() if dbg is not None else None
        if cfg.lwnemocou:
            assert wam2nemo is not None
        stream = torch.cuda.current_stream().cuda_stream
        rc = load().ecwam_v2_implsch(self.device_tables(), dp, int(kijs), int(kijl), fl1.data_ptr(), wvprpt.data_ptr(), ff.data_ptr(), intf.data_ptr(),
                                     mij.data_ptr(), xllws.data_ptr(), pw, pd, cfg.nang, cfg.nfre, variant_bits(cfg), stream)
        if rc:
            raise RuntimeError(f"ecwam_v2_implsch rc={rc}")
        self._test_last = 2

    def implsch_generation_used(self) -> int:
        return 2 if getattr(self, "_test_last", 0) == 2 else product_used(self)

    api.HipContext.set_implsch_generation = set_implsch_generation
    api.HipContext.implsch = implsch
    api.HipContext.implsch_generation_used = implsch_generation_used
    api.HipContext._v2_installed = True
