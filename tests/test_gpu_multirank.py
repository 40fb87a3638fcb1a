"""The N>1 code of bench.py on ONE GPU (-m gpu): two fresh child processes (torch.distributed.run, gloo) share device 0 and exchange the
advection halo host-staged through the library (ecwam_hip_halo_setup / _pack_host / _unpack_host); decomposition, overlap of the
exchange with the interior stencil, IMPLSCH on each band, the all_reduce of the timings -- everything bench.py --gpus N runs except
the RCCL transport itself, which needs one GPU per rank.  The spectra after a few steps equal the single-rank run bit for bit."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env):
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]      # ONE line on stdout: the JSON line (library chatter goes to stderr)
    return json.loads(lines[0])


@pytest.mark.parametrize("nranks,launcher,extra", [(2, "self", []), (3, "torchrun", []), (2, "self", ["--irefra", "2"]),
                                                   (3, "self", ["--ifrelfmax", "5", "--adv-per-source", "2"]),
                                                   (2, "self", ["--fused", "off"])])      # the two-kernel step (36 x 36 runs the one-kernel step by default)
# (Three ranks is what one card allows here: a GPU box admits 6 processes with the device open, and this process, the launcher and its
# agent hold it beside the ranks -- five ranks ended in the box's process guard.  The 8-rank decomposition a SCALE run executes is covered
# on the CPU: tests/test_host.py::test_eight_rank_decomposition_and_grid_file and the world-8 gloo exchange.)
def test_bench_two_ranks_on_one_gpu_match_single_rank(tmp_path, nranks, launcher, extra):
    """launcher "self": the driver's command shape, `python bench.py --gpus N ...` with WORLD_SIZE unset -- bench.py starts its own N
    ranks as a child process; "torchrun": started under torch.distributed.run as the contract's N > 1 command does."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    # extra = --ifrelfmax 5 --adv-per-source 2: the O1280 structure -- fast-wave sub-steps on compact rows (their own, smaller halo exchange),
    # the compact copy handed from the first advection step to the second.
    # extra = --irefra 2: currents; on N > 1 ranks the extended environment rows come from PROENVHALO on the device (owned rows + one
    # exchange of 3 NFRE + 3 reals per halo row through the same transport), on one rank from the host assembly: same bits
    common = ["--grid", "48", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"] + extra
    one = _run([sys.executable, "bench.py", "--gpus", "1", "--dump", str(tmp_path / "one")] + common, env)
    tail = ["bench.py", "--gpus", str(nranks), "--share-gpu", "--dump", str(tmp_path / "many")] + common
    if launcher == "self":
        many = _run([sys.executable] + tail, env)
    else:
        port = 29500 + (os.getpid() % 2000) + nranks
        many = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nranks}", "--master-addr", "127.0.0.1",
                     "--master-port", str(port)] + tail, env)
    assert many["n_gpus"] == nranks and many["config"]["ranks"] == nranks and many["config"]["halo"] == "host" and many["finite"] and one["finite"]
    a = np.load(str(tmp_path / "one") + ".0.npy")
    b = np.concatenate([np.load(str(tmp_path / "many") + f".{r}.npy") for r in range(nranks)])
    assert a.shape == b.shape and np.array_equal(a, b)
    assert many["value"] > 0 and abs(many["swh_norm_rank0"]["max"]) < 50


def test_bench_refuses_more_gpus_than_the_node_has():
    """`bench.py --gpus N` on a node with fewer than N devices must fail loudly, not run one rank (a one-GPU figure in an N-GPU record)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", str(n), "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and "visible" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_library_rccl_exchange_loopback_on_one_gpu(tmp_path):
    """The library's own MPEXCHNG transport executed on hardware with the one GPU there is: a communicator of ONE rank (RCCL refuses two
    ranks on one device) whose halo segment is fed by the rank itself -- dlopen of librccl, ncclGetUniqueId / ncclCommInitRank /
    ncclCommCount, the pack kernel on the caller's stream, the grouped ncclSend + ncclRecv on the library's stream behind an event, the
    receive landing in the halo rows, the caller's stream waiting at halo_finish, twice in a row without a host synchronisation (the
    second exchange reuses the send buffer behind the first one's event), on full rows and on compact fast-wave rows.  In a child
    process under a timeout: a transport that hangs must fail the test, not the session."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    code = r"""
import sys, types, numpy as np, torch
sys.path.insert(0, %r)
from ecwam_amd import api
from ecwam_amd.tables import Config, Tables
t = Tables(Config(nang=12, nfre=36, nfre_red=25), np.float32)
ctx = api.HipContext(t)
n, nh = 700, 60
rng = np.random.default_rng(3)
send = np.sort(rng.choice(n, nh, replace=False)).astype(np.int32)
dom = types.SimpleNamespace(rank=0, nranks=1, send={0: send}, recv={0: (n, nh)})
ctx.halo_setup(dom)
ctx.comm_init(ctx.comm_unique_id())
assert ctx.comm_count() == 1
for rowshape in ((12, 36), (12, 8)):
    fl = torch.zeros((n + nh + 1,) + rowshape, dtype=torch.float32, device=ctx.device)
    for it in range(2):
        fl[:n] = torch.from_numpy(rng.uniform(0, 1, (n,) + rowshape).astype(np.float32)).to(ctx.device)
        fl[n:n + nh] = -1.0
        ctx.halo_start(fl)
        fl[0:5] += 0.0                      # work on the caller's stream between start and finish
        ctx.halo_finish()
        got = fl[n:n + nh].clone()          # ordered behind the exchange on the caller's stream
        torch.cuda.synchronize()
        assert torch.equal(got, fl[torch.from_numpy(send).long().to(ctx.device)]), (rowshape, it)
        assert float(fl[n + nh].abs().max()) == 0.0
# two exchanges of different rows outstanding at once (compact fast-wave rows posted first, then the full rows: propag_wam.F90:166,293
# as WAMINTGR_HIP / Wamintgr._propag_fast_compact post them): each has its own send buffer and event pair, one finish waits for both
fa = torch.zeros((n + nh + 1, 12, 8), dtype=torch.float32, device=ctx.device)
fb = torch.zeros((n + nh + 1, 12, 36), dtype=torch.float32, device=ctx.device)
for it in range(3):
    fa[:n] = torch.from_numpy(rng.uniform(0, 1, (n, 12, 8)).astype(np.float32)).to(ctx.device)
    fb[:n] = torch.from_numpy(rng.uniform(0, 1, (n, 12, 36)).astype(np.float32)).to(ctx.device)
    fa[n:n + nh] = -1.0; fb[n:n + nh] = -1.0
    ctx.halo_start(fa)
    ctx.halo_start(fb)
    ctx.halo_finish()
    ga, gb = fa[n:n + nh].clone(), fb[n:n + nh].clone()
    torch.cuda.synchronize()
    ix = torch.from_numpy(send).long().to(ctx.device)
    assert torch.equal(ga, fa[ix]) and torch.equal(gb, fb[ix]), it
ctx.close()
print("loopback ok")
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert r.returncode == 0 and "loopback ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_halo_transports_agree_between_two_devices(tmp_path):
    """Runs where two GPUs are visible (skipped on the one-GPU boxes): bench.py --gpus 2 on two devices with the library's RCCL
    exchange (ecwam_hip_halo_start / _finish over xGMI: mpexchng.F90:141-246), with torch.distributed P2P and with the host-staged
    transport -- the same spectra bit for bit, `halo_rccl_ranks == 2` for the library transport, and no silent fallback (--strict-halo
    is the default without --share-gpu)."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    common = ["--grid", "48", "--steps", "3", "--warmup", "1", "--repeats", "1", "--no-cpu-baseline", "--ifrelfmax", "5", "--adv-per-source", "2"]
    outs = {}
    for halo in ("lib", "torch", "host"):
        outs[halo] = _run([sys.executable, "bench.py", "--gpus", "2", "--halo", halo, "--dump", str(tmp_path / halo)] + common, env)
        assert outs[halo]["config"]["halo"] == halo and outs[halo]["config"]["strict_halo"] is True and outs[halo]["finite"]
        assert len(outs[halo]["propag_split_per_rank"]) == 2
    assert outs["lib"]["config"]["halo_rccl_ranks"] == 2
    one = _run([sys.executable, "bench.py", "--gpus", "1", "--dump", str(tmp_path / "one")] + common, env)
    a = np.load(str(tmp_path / "one") + ".0.npy")
    for halo in ("lib", "torch", "host"):
        b = np.concatenate([np.load(str(tmp_path / halo) + f".{r}.npy") for r in range(2)])
        assert np.array_equal(a, b), halo


def test_strict_halo_is_the_default_and_fails_loudly(tmp_path):
    """A transport that cannot be set up must end `bench.py --gpus N` with a non-zero exit and no JSON line when --strict-halo is in force
    (forced here through bench.py's test hook ECWAM_BENCH_FAIL_HALO_SETUP); --no-strict-halo falls back and says so."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", ECWAM_BENCH_FAIL_HALO_SETUP="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    common = ["--grid", "48", "--steps", "1", "--warmup", "0", "--repeats", "1", "--no-cpu-baseline", "--share-gpu"]
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--strict-halo"] + common, cwd=ROOT, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], r.stdout[-1000:] + r.stderr[-2000:]
    assert "strict-halo" in r.stderr
    ok = _run([sys.executable, "bench.py", "--gpus", "2", "--no-strict-halo"] + common, env)
    assert ok["config"]["halo"].startswith("host (fallback") and ok["config"]["strict_halo"] is False


def test_bench_refuses_a_counter_summary_of_another_workload():
    """`bench.py --pmc-file` attaches HBM traffic and the busy fractions of a rocprofv3 counter summary to the roofline object only when the
    summary describes the run's own workload and kernel: the committed O320 summary handed to an O48 run must be refused loudly (no JSON line)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    pmc = os.path.join(ROOT, "profiles", "r05_bench_O320_sp_pmc.json")
    assert os.path.exists(pmc)
    r = subprocess.run([sys.executable, "bench.py", "--grid", "48", "--steps", "1", "--warmup", "0", "--repeats", "1", "--no-cpu-baseline", "--pmc-file", pmc],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and "refused" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], r.stderr[-400:]
