"""The known-answer checks of tests/known_answers.py on the DEVICE (-m gpu): the HIP library through the C ABI held directly to the
published equations -- Janssen's growth rate through the whole of IMPLSCH, the momentum balance of WNFLUXES, the f^-5 tail of IMPHFTAIL,
Kenyon's Stokes drift, the Courant numbers of the corner-transport upstream scheme -- without the oracle in between.  (The same checks run
on the oracle in tests/test_known_answers.py.)"""
import pytest

import known_answers as KA

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


@pytest.mark.parametrize("prec", ["dp", "sp"])
def test_device_wind_input_growth_rate_and_momentum_balance(prec):
    _need_gpu()
    run = KA.DeviceRun(prec)
    KA.check_momentum_balance(run, prec, KA.check_wind_input_growth_rate(run, prec))


@pytest.mark.parametrize("prec", ["dp", "sp"])
def test_device_tail_and_stokes_drift(prec):
    _need_gpu()
    KA.check_tail_and_stokes_drift(KA.DeviceRun(prec), prec)


@pytest.mark.parametrize("prec", ["dp", "sp"])
def test_device_ctu_single_bin(prec):
    _need_gpu()
    KA.check_ctu_single_bin(KA.DeviceRun(prec), prec)
