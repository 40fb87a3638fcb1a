import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# the oracle's OpenMP loops: the GPU box reports 256 logical CPUs and gives a job 16 of them; libgomp reads the variable when it loads
os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(16, len(os.sched_getaffinity(0))))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_built():
    from oracle import oracle

    oracle.build()
    return True


# Evidence runs with OTHER random inputs than the ones the gates were set on: ECWAM_TEST_SEED_OFFSET=<k> adds k to the seed of every
# synthetic sea state the tests generate (tests only: the patch lives here, ecwam_amd/synthetic.py is untouched).  Not part of the suite's
# contract -- the gates are observed maxima with a margin, on the default seeds; profiles/r05_seed_robustness.txt records what other seeds do.
_off = int(os.environ.get("ECWAM_TEST_SEED_OFFSET", "0") or 0)
if _off:
    from ecwam_amd import synthetic as _syn

    _point_params = _syn.point_params

    def _shifted(nglobal, seed=12345, *args, **kw):
        return _point_params(nglobal, seed + _off, *args, **kw)

    _syn.point_params = _shifted
