"""-m gpu: the N > 1 path of bench.py as it is launched by the driver -- one process per rank under torch.distributed.run, a
barrier on both sides of the timed region, MAX of the step time over the ranks, every rank's parity check ANDed -- started as a
CHILD process with two ranks.  On a box with >= 2 GPUs the ranks use DISTINCT devices and rendezvous over RCCL (the driver's own
launch); on the one-GPU test box both ranks use device 0 and rendezvous over gloo (RCCL wants one device per rank) -- everything else
is the code the 8-GPU run executes.  The JSON line names every rank's device (`ranks`), so a scaling record proves N distinct GPUs."""
import json
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ndev():
    # (torch's count does not initialise the GPU on this image)
    import torch
    return torch.cuda.device_count()


@pytest.fixture
def _bench(clean_launcher):
    """bench.py --gpus 2 started by the session's clean launcher (tests/conftest.py): a process that has touched the GPU must not
    fork + exec the ranks, and this one may have -- whatever ran before this file."""
    def run(*extra, gpus=2):
        return _bench_via(clean_launcher, *extra, gpus=gpus)
    return run


def _bench_via(launch, *extra, gpus=2):
    shared = ["--dist-backend", "gloo", "--all-ranks-on-device", "0"] if _ndev() < gpus else []       # enough GPUs: one device per rank, RCCL
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus)] + shared + ["--steps", "2",
           "--warmup", "1", "--configs", "none", "--no-extras", "--no-cpu-baseline", "--inflight", "1"] + list(extra)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    rc, out, err = launch(cmd, env=env, cwd=ROOT, timeout=900)
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert rc == 0 and len(lines) == 1, (rc, out[-2000:], err[-3000:])
    assert out.splitlines()[-1] == lines[0] and len(lines[0]) < 4096          # the contract line: small, and LAST
    return json.loads(lines[0])


def test_two_ranks_weak_scaling_every_rank_its_own_batch(_bench):
    d = _bench("--streams", "512", "--scaling", "weak")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 2
    assert d["config"]["parity_ok"] is True and d["config"]["verified_vs_oracle"] is True
    assert d["config"]["streams_this_rank"] == 512 and d["config"]["streams_whole_job"] == 1024
    assert d["value"] > 0 and d["roofline"]["frac"] > 0
    assert d["roofline"]["kernel_ms"] <= d["ms_per_step"] * 1.001            # both clocks around the same K launches
    # every rank reports the device it ran on
    assert [r["rank"] for r in d["ranks"]] == [0, 1] and all("gfx950" in r["device"] for r in d["ranks"])
    if _ndev() >= 2:
        assert len({(r.get("pci_domain_id"), r.get("pci_bus_id"), r.get("pci_device_id"), r.get("uuid")) for r in d["ranks"]}) == 2, d["ranks"]
        assert [r["local_rank"] for r in d["ranks"]] == [0, 1]


def test_two_ranks_strong_scaling_one_mixed_batch_partitioned_by_the_library(_bench):
    """BASELINE.json configs[3] in small: ONE mixed LZ10 / LZ11 / Yaz0 / PRS batch, alz_partition_batch decides which rank decodes what."""
    d = _bench("--streams", "512", "--scaling", "strong", "--format", "mixed")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["parity_ok"] is True
    assert d["config"]["streams_whole_job"] == 512 and 0 < d["config"]["streams_this_rank"] < 512


def test_two_ranks_compress_their_own_buffers(_bench):
    """BASELINE.json configs[4] ("LZSS compression ... 1 -> 8 GPUs scaling") as a two-rank job: device-resident encode per rank,
    round trip on the device, the first buffers byte for byte against the oracle's restatement of the managed encoder."""
    d = _bench("--mode", "encode", "--streams", "256", "--quality", "8")
    assert d["n_gpus"] == 2 and d["config"]["mode"] == "encode" and d["config"]["quality"] == 8
    assert d["config"]["parity_ok"] is True and d["config"]["verified_roundtrip_and_vs_oracle"] is True
    assert d["config"]["streams_whole_job"] == 512 and 0.1 < d["config"]["ratio"] < 0.6


def test_eight_ranks_strong_scaling_one_mixed_batch(_bench):
    """The launch the driver's SCALE run makes at N = 8, with no first-time code left in it: `bench.py --gpus 8` spawns eight rank processes under
    torch.distributed.run, eight contexts, ONE 2 048-stream mixed LZ10 / LZ11 / Yaz0 / PRS batch (BASELINE.json configs[3]) partitioned by
    alz_partition_batch into eight shards, a barrier either side of the timed region, MAX of the step time, the parity flag ANDed over all eight ranks,
    `ranks[]` with eight entries in rank order.  On a box with fewer than eight GPUs every rank uses device 0 and the rendezvous is gloo (RCCL wants a device
    per rank); with eight, the ranks use distinct devices over RCCL -- exactly the driver's launch.  No scaling number is expected of a one-GPU box."""
    d = _bench("--streams", "2048", "--scaling", "strong", "--format", "mixed", gpus=8)
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["steps"] == 2
    assert d["config"]["parity_ok"] is True and d["config"]["verified_vs_oracle"] is True
    assert d["config"]["streams_whole_job"] == 2048 and 0 < d["config"]["streams_this_rank"] < 2048
    assert [r["rank"] for r in d["ranks"]] == list(range(8)) and all("gfx950" in r["device"] for r in d["ranks"])
    assert d["value"] > 0 and d["roofline"]["kernel_ms"] <= d["ms_per_step"] * 1.001
    if _ndev() >= 8:
        assert [r["local_rank"] for r in d["ranks"]] == list(range(8))
        assert len({(r.get("pci_domain_id"), r.get("pci_bus_id"), r.get("pci_device_id"), r.get("uuid")) for r in d["ranks"]}) == 8, d["ranks"]


def test_eight_ranks_weak_scaling_yaz0(_bench):
    """... and the headline's own shape at N = 8 (weak scaling: every rank its own Yaz0 batch, the metric's `value` the sum over ranks / the slowest rank's time)."""
    d = _bench("--streams", "256", "--scaling", "weak", gpus=8)
    assert d["n_gpus"] == 8 and d["scaling"] == "weak"
    assert d["config"]["parity_ok"] is True and d["config"]["streams_this_rank"] == 256 and d["config"]["streams_whole_job"] == 2048
    assert [r["rank"] for r in d["ranks"]] == list(range(8))
