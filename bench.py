#!/usr/bin/env python3
"""bench.py -- grid-point spectral steps/sec of the WAMINTGR hot path on MI355X.

One "step" = one WAMINTGR cycle on synthetic forcing: advection (halo exchange + PROPAGS2) + NEWWIND +
IMPLSCH over every owned sea point, state resident in HBM.  Workload at N=1: octahedral
O320 all-ocean grid (421 080 sea points), 36 directions x 36 frequencies, single precision -- the
configuration BASELINE.json's metric is quoted on.  For N>1 (one process per GPU, launched by
torch.distributed.run) the per-GPU work is held fixed (weak scaling) on BASELINE.json's own grids: O453 at N=2, O640 at N=4,
O1280 at N=8 (configs 4 and 5; O1280 / 8 is 2x the per-GPU points of O320), sharded into contiguous sea-point ranges with a
point-to-point halo exchange over RCCL.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VALU_PEAK_TFLOPS = {"sp": 157.3, "dp": 78.6}      # MI355X_MICROARCH.md: vector (non-matrix) peaks
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def usable_cores() -> int:
    """Cores this process may actually run on: scheduler affinity, capped by the cgroup CPU quota (a GPU box hands one GPU's
    share of a large host to the job; os.cpu_count() reports the whole host)."""
    n = len(os.sched_getaffinity(0))
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(f) as fh:
                tok = fh.read().split()
            if f.endswith("cpu.max"):
                if tok[0] != "max":
                    n = min(n, max(1, int(float(tok[0]) / float(tok[1]) + 0.5)))
            else:
                q = int(tok[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                    per = int(fh.read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def self_launch(a, argv) -> None:
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): start the N ranks as a CHILD process -- one
    torch.distributed.run with N workers, rendezvous on 127.0.0.1 -- before this process has made any GPU call (never an exec
    after one), let its output through (rank 0 prints the JSON line) and exit with its code.  Fewer than N visible devices is an
    error, not a silent one-GPU run (torch.cuda.device_count() does not initialise the GPU)."""
    import socket
    import subprocess

    ndev = torch.cuda.device_count()
    if not a.share_gpu and ndev < a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but only {ndev} GPU(s) visible (use --share-gpu to rehearse the N>1 code on one GPU)")
    if a.share_gpu and ndev < 1:
        raise SystemExit("bench.py: --share-gpu needs one GPU")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    print(f"[bench] no launcher (WORLD_SIZE unset): starting {a.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    raise SystemExit(subprocess.run(cmd).returncode)


def cpu_baseline(nang: int, nfre: int, prec: str, target_s: float = 9.0, ng: int = 96) -> dict:
    """Oracle (plain-C restatement, OpenMP over points) timed on this host on a bounded sample of the same workload: the O96 all-ocean
    grid (40 280 sea points: 209 MB of spectra per copy in single precision, 1.7 GB of packed CTU weights -- beyond the last-level cache
    share of the cores used, so that the advection rate is a memory rate as it is at O320), the bench's precision first, then the
    other one on a shorter run."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from ecwam_amd import grid as G
    from ecwam_amd.tables import Config
    cores = int(os.environ.get("ECWAM_BENCH_CPU_THREADS", "0")) or usable_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)     # read by libgomp when the oracle library is loaded below
    from ecwam_amd import synthetic as syn
    from ecwam_amd.tables import Tables
    from oracle.oracle import Oracle

    g = G.build_grid(ng)
    n = g.nsea

    def leg(pr_: str, budget: float) -> dict:
        cfg = Config(nang=nang, nfre=nfre, nfre_red=nfre, idelt=450, idelpro=450)
        dt = np.float32 if pr_ == "sp" else np.float64
        t = Tables(cfg, dt)
        o = Oracle(cfg, pr_, fast=True)     # same sources, built -O3 -march=x86-64-v3 (oracle/Makefile): timing only
        p = syn.point_params(n)
        pr = syn.depth_props(p["DEPTH"], t, dt)
        fl = np.zeros((n + 1, nang, nfre), dt)
        fl[:n] = syn.jonswap_spectra(t.FR, t.TH, p["FP"], p["THETAQ"], dt)
        ff = syn.forcing(p, slice(0, n), t, dt)
        intf = np.zeros((n, syn.NINTF), dt)
        env = np.stack([pr["EMAXDPT"], p["DEPTH"].astype(dt)], 1)
        cg_ext = np.zeros((n + 1, nfre), dt)
        cg_ext[:n] = pr["CGROUP"]
        cg_ext[n] = syn.depth_props(np.array([998.999]), t, dt)["CGROUP"][0]
        w = o.ctu_weights(g, cg_ext, float(cfg.idelpro))
        # IMPLSCH through the NPROMA-blocked variant (SINPUT_ARD / SDISSIP_ARD / SNONLIN with the point index innermost, vector libm,
        # flush-to-zero: what the reference's loop order and production flags give a compiler), then a short run of the point-by-point
        # restatement for comparison; C calls only, arrays prepared once
        steps, el = o.timed_steps(g, fl.copy(), w, pr, env, ff.copy(), intf.copy(), max_steps=200, target_s=0.75 * budget)
        kind, t_impl, t_prop = o.implsch_kind, o.t_implsch, o.t_propags2
        s2, _ = o.timed_steps(g, fl, w, pr, env, ff, intf, max_steps=200, target_s=0.25 * budget, blocked=False)
        # which of the two is faster depends on the host (vector width, threads per core): the baseline is the step with the faster one
        per_blk, per_pt = t_impl / steps, o.t_implsch / s2
        per_step = el / steps - per_blk + min(per_blk, per_pt)
        if per_pt < per_blk:
            kind = "point-by-point"
        return {"value": n / per_step, "implsch_only": n / min(per_blk, per_pt), "propags2_only": n * steps / t_prop, "implsch_variant": kind,
                "implsch_only_nproma_blocked": n / per_blk, "implsch_only_point_by_point": n / per_pt, "steps": steps}

    main_leg = leg(prec, target_s)
    other = "dp" if prec == "sp" else "sp"
    other_leg = leg(other, 0.4 * target_s)
    steps = main_leg.pop("steps")
    out = {"value": main_leg.pop("value"), "unit": "grid-point spectral steps/s", "cores": cores, "kind": "port", **main_leg,
           other: {k: v for k, v in other_leg.items()},
           "note": "IMPLSCH is timed in two C restatements and the faster one on this host counts: NPROMA-blocked (oracle/ora_implsch_blk.inc: "
                   "SINPUT_ARD, SDISSIP_ARD and SNONLIN with the sea-point index innermost as in implsch.F90:152-170, omp simd + libmvec, "
                   "flush-to-zero; the scalar chains TAUT_Z0 / STRESSO / FKMEAN point by point) and point by point; PROPAGS2 its eight weights packed as contiguous streams (ora_propags2_w8, bit-identical); "
                   "both checked against the point-by-point oracle; unpinned against the reference "
                   "(DESIGN.md section 4): an estimate of what the reference's OpenMP path does on these cores, not a measurement of it",
           "sample": f"O{ng} all-ocean grid ({n} sea points, {n * nang * nfre * (4 if prec == 'sp' else 8) / 1e6:.0f} MB of spectra per copy: beyond the cores' cache), "
                     f"{nang}x{nfre} spectrum, {prec}, {steps} full steps "
                     f"(PROPAGS2 + IMPLSCH), oracle/ C restatement (speed build: gcc -O3 -march=x86-64-v3) with OpenMP over blocks of 32 points, "
                     f"{cores} threads (host reports {os.cpu_count()} logical CPUs); the other precision ('{other}') on the same sample, half the time"}
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed windows of --steps steps each, every one bracketed by barrier + synchronize; the MEDIAN window is reported "
                         "(SURVEY.md 8d: median of 5), min / max next to it")
    ap.add_argument("--grid", type=int, default=0, help="octahedral resolution (default 320*sqrt(gpus))")
    ap.add_argument("--prec", default="sp", choices=["sp", "dp"])
    ap.add_argument("--nang", type=int, default=36)
    ap.add_argument("--nfre", type=int, default=36)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-grid", type=int, default=0, help="octahedral grid of the CPU baseline sample: 320 = the benchmark's own grid (four "
                    "steps of ~2 s on 16 threads; 17.5 GB of packed CTU weights in single precision, 35 GB for the double precision leg), 96 = a "
                    "smaller sample beyond the cores' cache; default 0 = 320 where the host has more than 96 GB of memory available, else 96")
    ap.add_argument("--pmc-file", default="", help="counter summary of tools/pmc_bench.py to attach to the roofline object (HBM traffic, VALU / LDS busy "
                    "fractions); refused when its workload or kernel is not this run's.  Default: profiles/r05_bench_O320_sp_pmc.json where it matches")
    ap.add_argument("--weights", default="otf", choices=["otf", "stored"],
                    help="CTU weights rebuilt inside PROPAGS2 (default) or streamed from the stored W array")
    ap.add_argument("--strip", type=int, default=0, help="longitude-strip width of the advection work order (0: natural order)")
    ap.add_argument("--adv-per-source", type=int, default=1,
                    help="advection steps per source-term step (1: the O320 configuration; 2: O1280's native 450 s / 900 s ratio)")
    ap.add_argument("--ifrelfmax", type=int, default=0,
                    help="fast waves: frequencies 1..IFRELFMAX advected with two sub-steps of half the time step (O1280: 5)")
    ap.add_argument("--halo", default="lib", choices=["lib", "torch", "host"],
                    help="halo exchange (N > 1): the library's own MPEXCHNG (ecwam_hip_halo_start/_finish: grouped RCCL send/recv on the "
                         "library's stream -- what WAMINTGR_HIP calls; the default), torch.distributed P2P on packed buffers (RCCL), "
                         "or host staged through the CPU backend")
    ap.add_argument("--strict-halo", dest="strict_halo", action="store_true", default=None,
                    help="N > 1: a halo transport that cannot be set up or fails its self-check on any rank ends the run with a non-zero exit "
                         "instead of falling back to the host-staged transport (the default without --share-gpu: a scaling run must never "
                         "publish a host-staged curve under the name of the RCCL one)")
    ap.add_argument("--no-strict-halo", dest="strict_halo", action="store_false", help="allow the host-staged fallback (config.halo says so)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal of the N>1 code on ONE GPU: every rank uses device 0, process group on gloo, host-staged halo")
    ap.add_argument("--dump", default="", help="write the owned spectra of every rank to <path>.<rank>.npy after the last step")
    ap.add_argument("--host-state", action="store_true",
                    help="DIAGNOSTIC, never the reported value: the spectra enter from and return to pinned host memory every step (what a caller "
                         "that hands over host buffers would pay over PCIe; DESIGN.md section 6)")
    ap.add_argument("--fused", default="auto", choices=["auto", "off", "on"],
                    help="the 1:1 step as ONE kernel (ecwam_hip_propags2_implsch: PROPAGS2 inside IMPLSCH's tile load; bit-identical to the two "
                         "kernels).  auto (default): where a build covers the configuration (36 x 36, one advection step per source step, no "
                         "refraction / fast waves); on: refuse to run otherwise; off: PROPAGS2 and IMPLSCH as two kernels (the A/B partner)")
    ap.add_argument("--fused-flags", type=int, default=0, help="diagnostics: flags of ecwam_hip_propags2_implsch (1: natural workgroup order; 2: the probe)")
    ap.add_argument("--subgrid", action="store_true",
                    help="LSUBGRID (the reference's default on real bathymetry, mpuserin.F90:704): synthetic sub-grid obstruction coefficients scale the "
                         "space weights of the advection (ctuw.F90:703-733); not part of the BASELINE configurations (all-ocean grids)")
    ap.add_argument("--irefra", type=int, default=0, choices=[0, 1, 2, 3],
                    help="refraction: 0 none (BASELINE configurations), 1 depth, 2 currents, 3 both (synthetic current field)")
    a = ap.parse_args()

    if a.gpus < 1 or a.steps < 1 or a.repeats < 1:
        raise SystemExit("--gpus, --steps and --repeats must be >= 1")
    if a.strict_halo is None:
        a.strict_halo = not a.share_gpu
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a, sys.argv[1:])          # does not return
    # stdout carries ONE line, the JSON line of rank 0: whatever the libraries print there on any rank (Gloo announces its connections on
    # stdout) goes to stderr instead -- file descriptor 1 is pointed at stderr, the real stdout is kept for the result
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    from ecwam_amd import grid as G
    from ecwam_amd.tables import Config
    from ecwam_amd.wamintgr import Wamintgr
    # test hook (tests/test_gpu_multirank.py): the asked transport fails its set-up without RCCL being touched
    force_fail = bool(os.environ.get("ECWAM_BENCH_FAIL_HALO_SETUP"))
    if a.share_gpu:
        local_rank, a.halo = 0, ("lib" if force_fail else "host")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if a.share_gpu:
            dist.init_process_group("gloo")
        else:
            # "cpu:gloo,cuda:nccl": the host-staged halo (chosen, or the fallback of the self-check below) moves CPU tensors,
            # everything else goes over RCCL
            dist.init_process_group("cpu:gloo,cuda:nccl", device_id=torch.device("cuda", local_rank))

    ng = a.grid or {1: 320, 2: 453, 4: 640, 8: 1280}.get(world, int(round(320 * math.sqrt(world))))
    # time step: 450 s at O320 (the 900 s of the reference's 24-direction O320 yml violates the CTU stability criterion
    # with 36 directions near the poles of the all-ocean grid, ctuw.F90:637); scaled with the grid spacing beyond
    dt = 450 if ng <= 320 else max(15, int(450 * 320 / ng) // 15 * 15)
    cfg = Config(nang=a.nang, nfre=a.nfre, nfre_red=a.nfre, idelt=dt, idelpro=dt, irefra=a.irefra)
    # N > 1: rank 0 builds the grid tables once and hands them to the other ranks through a file (all ranks of a node building the whole
    # O1280 grid side by side cost 54 s of numpy each on the node's CPU share before the first step)
    if world > 1:
        import tempfile
        # (named by the rendezvous port: the ranks of one run share it whatever started them, two runs on a node cannot)
        gpath = os.path.join(tempfile.gettempdir(), f"ecwam_amd_grid_O{ng}_w{world}_{os.environ.get('MASTER_PORT', '0')}.npz")
        if rank == 0:
            grid = G.build_grid(ng)
            G.save_grid(grid, gpath)
        flag = torch.zeros(1, dtype=torch.int32)          # (a CPU tensor: reduced over gloo in either process group)
        dist.all_reduce(flag)
        if rank != 0:
            grid = G.load_grid(gpath)
        dist.all_reduce(flag)
        if rank == 0:
            os.remove(gpath)
    else:
        grid = G.build_grid(ng)
    # N > 1: the model starts on the host-staged transport (no RCCL involved: it cannot fail to set up), then moves to the asked one
    m = Wamintgr(cfg, grid, a.prec, device=local_rank, rank=rank, nranks=world, weights=a.weights, strip_width=a.strip,
                 ifrelfmax=a.ifrelfmax, delpro_lf=(dt / 2.0 if a.ifrelfmax else None), halo_transport="host" if world > 1 else "torch")
    m.init_synthetic(env_on_device=bool(a.irefra) and world > 1)      # refraction on N > 1 ranks: PROENVHALO on the device, halo rows through the transport
    m.ff_next = m.ff.clone()      # NEWWIND hands the (unchanged synthetic) forcing over every step: k_newwind is part of the step
    if a.subgrid:
        from ecwam_amd import synthetic as syn
        m.set_obstructions(syn.obstructions(grid, a.nfre, seed=5))
    nfail = m.build_weights()
    if dist is not None:      # every rank leaves together (the reference aborts the whole run, ctuwdrv.F90:124-146): nobody waits in a collective
        tot = torch.tensor([nfail], dtype=torch.int64)
        dist.all_reduce(tot)
        nfail = int(tot.item())
    if nfail:
        raise SystemExit(f"CFL violated at {nfail} points")

    # ---- N > 1: set the asked halo transport up and check it once before anything is timed -- every rank fills its owned rows with
    #      the global point index, exchanges, and compares its halo rows with the indices its neighbours own.  A transport that
    #      raises or delivers something else on ANY rank (first run on a new node: RCCL communicator / point-to-point set-up) is
    #      replaced by the host-staged one on all ranks, and the JSON line says so.
    halo_used, halo_ranks = a.halo, None
    if world > 1:
        from ecwam_amd.wamintgr import HaloExchange

        def all_ok(ok: bool) -> bool:
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32)          # CPU tensor: reduced over gloo
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return bool(int(flag.item()))

        def halo_ok() -> bool:
            try:
                t = torch.zeros((m.dom.nrows, 1, 4), dtype=m.fl1.dtype, device=m.fl1.device)
                t[: m.n, 0, 0] = torch.arange(m.dom.lo, m.dom.hi, dtype=torch.float64, device=t.device).to(t.dtype) % 65536.0
                m.halo(t)
                torch.cuda.synchronize()
                want = torch.from_numpy(np.asarray(m.dom.halo_global, dtype=np.float64) % 65536.0).to(t.dtype)
                return bool(torch.equal(t[m.n: m.n + m.dom.nh, 0, 0].cpu(), want))
            except Exception as e:          # noqa: BLE001 -- any failure of the transport means: use the other one
                print(f"[bench] rank {rank}: halo transport '{m.halo.transport}' failed its self-check: {e!r}", file=sys.stderr, flush=True)
                return False

        host_halo = m.halo
        if a.halo != "host":
            try:
                if force_fail:
                    raise RuntimeError("ECWAM_BENCH_FAIL_HALO_SETUP is set (test hook)")
                m.halo = HaloExchange(m.dom, m.dev, m.ctx, transport=a.halo)      # "lib": ncclCommInitRank inside the library
                ok = True
            except Exception as e:          # noqa: BLE001
                print(f"[bench] rank {rank}: halo transport '{a.halo}' could not be set up: {e!r}", file=sys.stderr, flush=True)
                ok = False
            if not all_ok(ok) or not all_ok(halo_ok()):
                if a.strict_halo:
                    raise SystemExit(f"bench.py: halo transport '{a.halo}' failed its set-up or its self-check on at least one rank "
                                     f"(--strict-halo; --no-strict-halo falls back to the host-staged transport)")
                m.halo = host_halo
                m.ctx.halo_setup(m.dom)
                halo_used = f"host (fallback: '{a.halo}' failed its set-up or self-check)"
        if isinstance(halo_used, str) and halo_used.startswith("host") and not all_ok(halo_ok()):
            raise SystemExit("halo exchange self-check failed")
        if m.halo.transport == "lib":
            halo_ranks = m.ctx.comm_count()       # what RCCL itself says (ncclCommCount)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # (several advection steps per source step: the LAST one is the tile load of the source-term kernel, the others run PROPAGS2)
    fused = a.fused != "off" and not a.host_state and a.weights == "otf" and m.fused_available()
    if a.fused == "on" and a.host_state:
        raise SystemExit("bench.py: --fused and --host-state exclude each other")
    if a.fused == "on" and not fused:
        raise SystemExit("bench.py: --fused on, but no one-kernel build covers this configuration")

    def step_untimed():
        if fused:
            for _ in range(a.adv_per_source - 1):
                m.propag()
            m.step_fused(flags=a.fused_flags)
            return
        for _ in range(a.adv_per_source):
            m.propag()
        m.newwind()
        m.implsch()

    host_fl = torch.empty((m.n, a.nang, a.nfre), dtype=m.fl1.dtype).pin_memory() if a.host_state else None
    if host_fl is not None:
        host_fl.copy_(m.fl1[: m.n])
    for _ in range(a.warmup):
        step_untimed()
    # a.repeats windows of a.steps steps, each bracketed by barrier + synchronize on both sides; a window's time is the MAX over ranks,
    # the reported one the median window.  Kernel times: HIP events on the launch stream around PROPAG_WAM and IMPLSCH of every step.
    m.halo_events = [] if world > 1 else None      # N > 1: the compute stream's wait at the end of every exchange (Wamintgr.propag)
    windows, kt = [], []
    for _ in range(a.repeats):
        sync()
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(a.steps)]
        if m.halo_events is not None:
            m.halo_events.clear()
        t0 = time.perf_counter()
        for s in range(a.steps):
            e = ev[s]
            if host_fl is not None:
                m.fl1[: m.n].copy_(host_fl, non_blocking=True)
                m.gfast_valid = False
            e[0].record()
            if fused:             # (the advection steps but the last,) NEWWIND, then halo exchange (N > 1) + the one kernel that advects and integrates
                for _ in range(a.adv_per_source - 1):
                    m.propag()
                e[1].record()
                e[2].record()
                m.step_fused(flags=a.fused_flags)
                e[3].record()
                continue
            for _ in range(a.adv_per_source):
                m.propag()        # halo exchange (N > 1) + PROPAGS2 (+ fast-wave sub-steps)
            e[1].record()
            m.newwind()
            e[2].record()
            m.implsch()
            e[3].record()
            if host_fl is not None:
                host_fl.copy_(m.fl1[: m.n], non_blocking=True)
        sync()
        windows.append(time.perf_counter() - t0)
        hw = sum(x.elapsed_time(y) for x, y in m.halo_events) / a.steps if m.halo_events else 0.0
        kt.append((sum(e[0].elapsed_time(e[1]) for e in ev) / a.steps, sum(e[2].elapsed_time(e[3]) for e in ev) / a.steps, hw))
    m.halo_events = None
    per_rank = None
    if dist is not None:
        dev = "cpu" if a.share_gpu else "cuda"
        tt = torch.tensor(windows, dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        windows = [float(x) for x in tt.cpu()]
    order_w = sorted(range(a.repeats), key=lambda i: windows[i])
    imed = order_w[(a.repeats - 1) // 2]        # the median window (the lower one of an even count)
    el = windows[imed]
    t_prop, t_impl, t_wait = kt[imed]
    if dist is not None:
        mine = torch.tensor([t_prop, t_impl, t_wait], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [[float(v) for v in x.cpu()] for x in allr]
        t_prop, t_impl = max(r_[0] for r_ in per_rank), max(r_[1] for r_ in per_rank)
    if a.dump:
        np.save(f"{a.dump}.{rank}.npy", m.fl1[: m.n].cpu().numpy())
    swh = m.swh()
    finite = bool(torch.isfinite(swh).all().item())
    swh_avg, swh_min, swh_max, _ = m.swh_norm()      # OUTWNORM of the significant wave height, computed on the device

    if rank == 0:
        w = 4 if a.prec == "sp" else 8
        N, NR = a.nang * a.nfre, a.nang * cfg.nfre_red
        b_impl = w * (3 * N + 5 * a.nfre + 55)          # F r+w, XLLWS w, 5 per-frequency props, ~55 scalars
        if a.irefra:
            b_prop = w * (2 * NR + 3 * a.nfre + 2 * a.nang + 18) + 60   # + own OMOSNH2KD / WAVNUM rows and the REFR row
        elif a.weights == "stored":
            b_prop = w * 10 * NR + 56                    # 8 weights + F1 + F3, 14 int32 neighbour ids
        else:
            b_prop = w * (2 * NR + a.nfre + 13) + 60     # F1 + F3 + own CGROUP row + point geometry, 15 int32 ids
        # vector flops per point-step (IMPLSCH 0.47 Mflop at 36 x 36; the advection: 61 per spectral bin = 8 multiply-adds of the
        # stencil + the rebuild of the eight weights in their hoisted form, csrc/ctu.h) against the 157.3 TFLOP/s fp32 vector peak (78.6 fp64)
        fl_impl = 0.47e6 * (a.nang * a.nfre) / 1296.0
        fl_prop = 61.0 * NR
        if fused:
            # ONE kernel per step: F1 read, F3 written, XLLWS written, the five per-frequency properties, the own CGROUP row and point geometry
            # of the weights, ~68 scalars, 15 int32 neighbour ids -- the advected spectrum never goes to memory (2 N w bytes less than the
            # two kernels together)
            b_step = w * (3 * N + 6 * a.nfre + 68) + 60
            kern = {"implsch_adv": {"ms": t_impl, "alg_bytes": b_step * m.n, "gbs": b_step * m.n / t_impl / 1e6, "flop": (fl_impl + fl_prop) * m.n,
                                    "what": "k_implsch4<..., ADV = 1 | 3>: PROPAGS2 inside IMPLSCH's tile load (+ k_ctu_prep, k_implsch4_pre / _fin"
                                            + (", + the fast waves' sub-steps on compact rows" if a.ifrelfmax else "") + ")"}}
            if a.adv_per_source > 1:      # the advection steps before the last one: PROPAGS2 on its own
                kern["propags2"] = {"ms": t_prop, "alg_bytes": b_prop * m.n * (a.adv_per_source - 1),
                                    "gbs": b_prop * m.n * (a.adv_per_source - 1) / t_prop / 1e6, "flop": fl_prop * m.n * (a.adv_per_source - 1)}
        else:
            kern = {
                "propags2": {"ms": t_prop, "alg_bytes": b_prop * m.n * a.adv_per_source,
                             "gbs": b_prop * m.n * a.adv_per_source / t_prop / 1e6, "flop": fl_prop * m.n * a.adv_per_source},
                "implsch": {"ms": t_impl, "alg_bytes": b_impl * m.n, "gbs": b_impl * m.n / t_impl / 1e6, "flop": fl_impl * m.n},
            }
        dom = max(kern, key=lambda k: kern[k]["ms"])
        # HBM traffic and the busy fractions of the vector ALU and the LDS are NOT measured in this run: they come from a counter summary
        # tools/pmc_bench.py took with rocprofv3 --pmc on this very command (profiles/r05_bench_O320_sp_pmc.json: separate passes for the SQ
        # sets, FETCH_SIZE and WRITE_SIZE; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, calibrated on a copy in
        # profiles/r02_fetch_size_calibration.json).  The summary is attached only when it describes THIS workload and THIS kernel: its workload
        # string and dtype must equal the run's and its kernel must be the instantiation the run launches; a file given with --pmc-file that
        # does not match is refused.
        workload = (f"O{ng} all-ocean octahedral grid, {grid.nsea} sea points, {a.nang} dir x {a.nfre} freq "
                    f"(NFRE_RED={cfg.nfre_red}), full WAMINTGR step = PROPAGS2 advection + NEWWIND + IMPLSCH"
                    + (" as one kernel (PROPAGS2 inside IMPLSCH's tile load)" if fused else "") + ", "
                    f"IDELT=IDELPRO={dt} s, flag set A (IPHYS=1, ISNONLIN=0, LLGCBZ0=F, LLNORMAGAM=F)"
                    + (f", IREFRA={a.irefra} (synthetic currents)" if a.irefra else "")
                    + (", LSUBGRID (synthetic obstruction coefficients)" if a.subgrid else "")
                    + (f", {a.adv_per_source} advection steps per source step" if a.adv_per_source != 1 else "")
                    + (f", fast waves M<={a.ifrelfmax} in two sub-steps" if a.ifrelfmax else "")
                    + (", DIAGNOSTIC: spectra copied from and to pinned host memory every step" if a.host_state else ""))
        dtype = "f32" if a.prec == "sp" else "f64"
        here = os.path.dirname(os.path.abspath(__file__))
        pmc_path = a.pmc_file or os.path.join(here, "profiles", f"r06_bench_O{ng}_{a.prec}_pmc.json")
        pmc, pmc_why = None, None
        from ecwam_amd import lib as L
        lib_file = L.LIBPATH      # (ECWAM_HIP_LIB, or the product library)
        lib_sha = hashlib.sha256(open(lib_file, "rb").read()).hexdigest()
        if os.path.exists(pmc_path) and world == 1:
            with open(pmc_path) as fh:
                cand = json.load(fh)
            ck = "implsch" if dom.startswith("implsch") else dom
            kname = (cand.get("kernels", {}).get(ck, {}) or {}).get("name") or ""
            tn = "float" if a.prec == "sp" else "double"
            want_k = {"implsch": f"k_implsch4<{tn}, {a.nang}, ", "propags2": "k_propags2"}[ck]
            adv_ok = (kname.rstrip(">").split(",")[-1].strip() == ("1" if fused else "0")) if ck == "implsch" else True
            if cand.get("workload", {}).get("workload") != workload or cand.get("workload", {}).get("dtype") != dtype:
                pmc_why = "the summary was taken on another workload: " + str(cand.get("workload", {}).get("workload"))
            elif want_k not in kname or not adv_ok:
                pmc_why = f"the summary's kernel is {kname!r}, this run launches {want_k}...>" + (" with the advecting tile load" if fused else "")
            elif cand.get("library", {}).get("sha256") != lib_sha:
                pmc_why = (f"the summary was taken on another build of the library (sha256 {str(cand.get('library', {}).get('sha256'))[:16]}..., "
                           f"this run loaded {lib_sha[:16]}...)")
            else:
                pmc = cand["kernels"][ck]
        elif a.pmc_file:
            pmc_why = "no such file" if not os.path.exists(pmc_path) else "counter summaries describe one GPU"
        elif world == 1:
            pmc_why = "no counter summary for this workload under profiles/"
        if a.pmc_file and pmc is None:
            raise SystemExit(f"bench.py: --pmc-file {a.pmc_file} refused: {pmc_why}")
        vec_peak = VALU_PEAK_TFLOPS[a.prec]
        roof = {"bound": "hbm", "kernel": dom, "achieved": kern[dom]["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": kern[dom]["gbs"] / HBM_PEAK_GBS,
                "traffic": None,
                # the roof that binds this kernel (above the fp32 ridge): its vector flops against the vector peak
                "valu_frac": kern[dom]["flop"] / (kern[dom]["ms"] * 1e-3) / (vec_peak * 1e12), "valu_peak_tflops": vec_peak,
                "valu_busy": None, "lds_busy": None, "counters": None, "counters_why_null": pmc_why, "library_sha256": lib_sha}
        if pmc is not None:
            # one story: the fraction of the HBM roofline on algorithmic bytes (the contract's figure), the bytes the kernel really moved,
            # and what its own counters say limits it -- the busy fractions of a SIMD's vector ALU and of the CU's LDS array at the resident
            # wave count (IMPLSCH: two waves per SIMD in single precision).  `bound` follows the counters: "hbm" only for a kernel that moves
            # its bytes at more than half the peak, else the busiest unit
            vb, lb = pmc.get("valu_busy"), pmc.get("lds_busy")
            hbm_frac_real = (pmc.get("hbm_bytes") or 0.0) / (kern[dom]["ms"] * 1e-3) / (HBM_PEAK_GBS * 1e9)
            bound = "hbm" if hbm_frac_real > 0.5 else ("valu+lds" if (vb or 0) > 0.5 and (lb or 0) > 0.4 else ("valu" if (vb or 0) > 0.5 else "latency"))
            roof.update({"bound": bound, "traffic": pmc.get("hbm_bytes"), "valu_busy": vb, "lds_busy": lb, "hbm_frac_of_moved_bytes": hbm_frac_real,
                         "limiter": "vector ALU + LDS at the resident wave count, not HBM" if dom.startswith("implsch") else "gather latency / instruction issue",
                         "counters": {"file": os.path.relpath(pmc_path, here), "kernel": pmc.get("name"), "library_sha256": lib_sha,
                                      "not_measured_live": True, "valu_insts_per_point": pmc.get("per_point", {}).get("SQ_INSTS_VALU"),
                                      "waitcnt_fraction": pmc.get("waitcnt_fraction"), "lds_bank_conflict_share": pmc.get("lds_bank_conflict_share")}})
            roof.pop("counters_why_null")
        out = {
            "metric": f"grid-point spectral steps/sec (whole node) at O{ng}, {a.nang}dir x {a.nfre}freq",
            "value": grid.nsea * a.steps / el, "unit": "grid-point spectral steps/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": el / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "repeats": a.repeats, "window_ms": {"median": el * 1e3, "min": min(windows) * 1e3, "max": max(windows) * 1e3,
                                                "all": [w_ * 1e3 for w_ in windows]},
            "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": workload,
                       "points_per_gpu": m.n, "halo_points": m.dom.nh, "parallelism": f"sea-point block x{world}",
                       "halo": halo_used if world > 1 else None, "halo_rccl_ranks": halo_ranks,
                       "strict_halo": bool(a.strict_halo) if world > 1 else None,
                       "grid_note": (f"O{ng} is not a BASELINE.json grid: weak scaling holds the points per GPU of O320 at N = {world}"
                                     if ng not in (48, 320, 640, 1280) else None),
                       "time_step_note": f"IDELT = IDELPRO = {dt} s: the reference's O320 yml runs 900 s with 24 directions "
                                         "(tests/etopo1_oper_an_fc_O320.yml:6-9); with 36 directions 900 s fails the CTU stability check "
                                         "near the poles of the all-ocean grid (ctuw.F90:637), 450 s passes; scaled with the grid spacing "
                                         "beyond O320.  The work per step does not depend on the time step",
                       "ranks": world, "share_gpu": bool(a.share_gpu)},
            # N > 1, per rank [PROPAG_WAM ms, IMPLSCH ms, of PROPAG_WAM: the compute stream waiting for the halo exchange at halo_finish]:
            # the exchange is posted first and the rows that read no halo row (decomp.interior) are advected while it runs, so the wait
            # is what the overlap did not hide
            "propag_split_per_rank": ([{"rank": i, "propag_ms": r_[0], "implsch_ms": r_[1], "halo_wait_ms": r_[2], "stencil_ms": r_[0] - r_[2]}
                                       for i, r_ in enumerate(per_rank)] if per_rank else None),
            "roofline": roof,
            "kernels": kern,
            "finite": finite,
            "swh_norm_rank0": {"avg": swh_avg, "min": swh_min, "max": swh_max},
        }
        if world == 1 and not a.no_cpu_baseline:
            cng = a.cpu_baseline_grid
            if cng <= 0:
                try:
                    import psutil
                    cng = 320 if psutil.virtual_memory().available > 96e9 else 96
                except Exception:
                    cng = 96
            try:
                out["cpu_baseline"] = cpu_baseline(a.nang, a.nfre, a.prec, ng=cng)
            except MemoryError:      # (the stored weights of the O320 sample did not fit after all)
                out["cpu_baseline"] = cpu_baseline(a.nang, a.nfre, a.prec, ng=96)
        print(json.dumps(out), file=result_out, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
