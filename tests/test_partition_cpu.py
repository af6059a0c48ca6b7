"""alz_partition_batch (host code of the multi-GPU path): greedy LPT over Sum D_i x per-format cost."""
import numpy as np

from auroralib.compression_amd import _abi as A
from auroralib.compression_amd.batch import partition_batch


def _streams(fmts, sizes):
    n = len(fmts)
    st = (A.Stream * n)()
    off = 0
    for i in range(n):
        st[i] = A.Stream(0, off, 100, sizes[i], sizes[i], 0, 0, fmts[i])
        off += sizes[i]
    return st


def test_equal_jobs_are_dealt_evenly():
    st = _streams([A.FMT_YAZ0] * 4000, [262144] * 4000)
    part, cost = partition_batch(st, 8)
    assert np.bincount(part, minlength=8).tolist() == [500] * 8
    assert cost.max() == cost.min()


def test_mixed_batch_is_balanced_by_cost_not_by_count():
    fm = [[A.FMT_LZ10, A.FMT_LZ11, A.FMT_YAZ0, A.FMT_PRS_BE][i % 4] for i in range(4000)]
    rng = np.random.default_rng(7)
    sizes = rng.integers(1000, 300000, size=4000).tolist()
    st = _streams(fm, sizes)
    part, cost = partition_batch(st, 8)
    assert set(part.tolist()) == set(range(8))
    assert (cost.max() - cost.min()) / cost.mean() < 0.01           # LPT: within one (small) job of each other
    # PRS streams cost about three Yaz0 streams of the same size: a part made of PRS alone holds fewer bytes
    bytes_per_part = np.bincount(part, weights=np.array(sizes, dtype=np.float64), minlength=8)
    assert bytes_per_part.max() / bytes_per_part.min() < 1.5


def test_one_huge_stream_gets_a_part_of_its_own():
    st = _streams([A.FMT_YAZ0] * 9, [10_000_000] + [1000] * 8)
    part, cost = partition_batch(st, 2)
    assert (part[1:] != part[0]).all()


def test_degenerate_inputs():
    part, cost = partition_batch(_streams([], []), 4)
    assert len(part) == 0 and cost.tolist() == [0, 0, 0, 0]
    part, cost = partition_batch(_streams([A.FMT_LZ10], [5]), 3)
    assert part.tolist() == [0]
