"""A fixed slice of the randomised differential run (tools/fuzz_parity.py) inside the GPU suite: random class, geometry (odd sizes, one-row
and one-column frames), parameters, number of streams, entry point (host frames, device batches, clips, a mix), independent stream resets -
every mask, every background and the model at the end against the oracle.  Seeds are fixed, so a failure names the case that reproduces
it: `python tools/fuzz_parity.py 1 <seed> v`.  The open-ended run (thousands of cases, profiles/r04_fuzz_parity.txt) stays a tool."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("first,count,big", [(41000, 300, False), (52000, 200, False), (63000, 80, True)])
def test_fixed_slice_of_the_fuzz_run(first, count, big):
    from tools import fuzz_parity
    fuzz_parity.BIG = big
    fuzz_parity.VERBOSE = False
    try:
        ran = 0
        for seed in range(first, first + count):
            msg = fuzz_parity.one_case(np.random.default_rng(seed), seed)
            ran += msg.endswith(": ok")
        assert ran >= count // 2, "most cases must be accepted by the engine (%d of %d ran)" % (ran, count)
    finally:
        fuzz_parity.BIG = False
