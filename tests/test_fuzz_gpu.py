"""-m gpu: a bounded run of the differential fuzzer (tests/fuzz_gpu.py) inside the collected suite: 24 random ragged
batches — sizes 1..9000, continuous / gridded / int16 / float64 data, both test masks, random window and method —
through the HIP path against the oracle.  `python tests/fuzz_gpu.py 400` runs it for longer."""
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
@pytest.mark.parametrize('seed', [12345, 20261002])
def test_bounded_differential_fuzz(seed):
    import fuzz_gpu
    seen = []
    fuzz_gpu.run(12, seed, log=seen.append)
    assert len(seen) == 12
