#!/usr/bin/env python3
"""ISA summary of the IMPLSCH kernels (diagnostic): compiles a translation unit to gfx950 assembly with the product flags and prints,
per kernel asked for, the resources (VGPR / AGPR / SGPR / scratch / occupancy; the LDS is dynamic: see implsch_v4_launch.h) and an
instruction-class histogram of the whole kernel and of every loop in it (static counts: one pass of a loop body).

python tools/isa_summary.py [source.hip] [substring of the mangled kernel name ...]  > profiles/r03_implsch4_isa_summary.txt
default: implsch4.hip, the three builds of the benchmark configuration (k_implsch4<float,36,3,1,3,8,false>, _pre, _fin).
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ecwam_amd import build as B  # noqa: E402

TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def classify(op: str) -> str:
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm", "s_barrier")):
        return "branch / end"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith(("s_load", "s_buffer_load")):
        return "SMEM (scalar loads)"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("ds_bpermute") or op.startswith("ds_permute"):
        return "LDS crossbar (ds_bpermute)"
    if op.startswith("ds_"):
        return "LDS read / write"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM (global loads / stores)"
    if op.startswith("v_accvgpr"):
        return "VALU accvgpr moves"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "VALU lane <-> SGPR"
    if op.endswith("_dpp") or "_dpp" in op:
        return "VALU DPP"
    if op.startswith("v_pk_"):
        return "VALU packed (v_pk_*)"
    if op.startswith(TRANS):
        return "VALU transcendental"
    if op.startswith("v_"):
        return "VALU other"
    return "other"


def summarise(lines):
    h = collections.Counter()
    for ln in lines:
        t = ln.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        h[classify(t.split()[0])] += 1
    return h


def main() -> None:
    args = sys.argv[1:]
    src = next((a for a in args if a.endswith(".hip")), "implsch4.hip")
    want = [a for a in args if not a.endswith(".hip")] or ["k_implsch4IfLi36ELi3ELi1ELi3ELi8ELb0E", "k_implsch4_preIfLb0E", "k_implsch4_finIfLb0E"]
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        variant = os.environ.get("ISA_VARIANT", "")      # a key of ecwam_amd.build.VARIANTS
        flags = B.FLAGS + (B.VARIANTS[variant] if src in B.IMPLSCH_SOURCES else [])
        cmd = [B.HIPCC, *[f for f in flags if f != "-fPIC"], "-S", "--cuda-device-only", "-o", out, os.path.join(B.CSRC, src)]
        subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        text = open(out).read().split("\n")
    print(f"# {src}: hipcc {' '.join(f for f in flags if f != '-fPIC')} -S --cuda-device-only   (static instruction counts)")
    starts = [(i, ln.split(":")[0]) for i, ln in enumerate(text) if re.match(r"^_Z\w+:", ln)]
    for w in want:
        for n, (i, name) in enumerate(starts):
            if w not in name:
                continue
            end = starts[n + 1][0] if n + 1 < len(starts) else len(text)
            body = text[i:end]
            stop = next((k for k, ln in enumerate(body) if ln.strip().startswith("s_endpgm")), len(body))
            code = body[: stop + 1]
            print(f"\n## {name}")
            res = {}
            for ln in text:
                m = re.match(r"\s*\.set " + re.escape(name) + r"\.(\w+), (\S+)", ln)
                if m:
                    res[m.group(1)] = m.group(2)
            meta = [ln.strip() for ln in body[stop:] if re.match(r"\s*; (NumVgprs|NumAgprs|TotalNumVgprs|TotalNumSgprs|ScratchSize|Occupancy|LDSByteSize)", ln)]
            print("resources: " + "; ".join(m.lstrip("; ") for m in meta))
            tot = summarise(code)
            print(f"whole kernel ({sum(tot.values())} instructions): " + ", ".join(f"{k} {v}" for k, v in tot.most_common()))
            # loops: a label carrying "Loop Header" up to the backward branch to it
            labels = {ln.split(":")[0]: k for k, ln in enumerate(code) if re.match(r"^\.LBB\d+_\d+:", ln)}
            for lab, k0 in sorted(labels.items(), key=lambda kv: kv[1]):
                if "Loop Header" not in code[k0]:
                    continue
                k1 = max((k for k, ln in enumerate(code) if k > k0 and re.search(r"s_c?branch\w*\s+" + re.escape(lab) + r"\b", ln)), default=None)
                if k1 is None:
                    continue
                h = summarise(code[k0:k1 + 1])
                valu = sum(v for c, v in h.items() if c.startswith("VALU"))
                depth = re.search(r"Depth=(\d+)", code[k0])
                print(f"  loop {lab} (depth {depth.group(1) if depth else '?'}, {sum(h.values())} instructions, {valu} VALU): "
                      + ", ".join(f"{c} {v}" for c, v in h.most_common()))


if __name__ == "__main__":
    main()
