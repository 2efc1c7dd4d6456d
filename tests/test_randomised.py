"""A fixed-seed slice of the two randomised sweeps (tests/fuzz_parity.py, tests/sweep_qhull_device.py: the
round functions themselves, imported) so that every run of the GPU tests carries them: kernel shape boundaries,
degenerate content, binning edge cases and the whole hybrid chain against the CPU restatements, and the device
replay of Qhull's construction against the host replay.  Needs a GPU; the long runs stay a job done by hand."""
import pytest

pytestmark = pytest.mark.gpu


def test_forty_rounds_of_the_parity_sweep():
    import fuzz_parity as Z
    del Z.bad[:]
    slices, points = Z.sweep(40, 31000)
    assert slices > 5000 and points > 1_000_000
    assert Z.bad == []


def test_rounds_of_the_device_replay_of_qhulls_order_against_the_host_replay():
    import sweep_qhull_device as Q
    bad, open_, slices = Q.main(rounds=8, seed=31)
    assert slices > 1000
    assert bad == 0 and open_ == 0
