"""-m gpu: the N > 1 path of bench.py as it is launched by the driver -- one process per rank under torch.distributed.run, a
barrier on both sides of the timed region, MAX of the step time over the ranks, every rank's parity check ANDed -- started as a
CHILD process with two ranks.  The test box has one GPU, so both ranks use device 0 and rendezvous over gloo (RCCL wants one device
per rank); everything else is the code the 8-GPU run executes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--all-ranks-on-device", "0", "--steps", "2",
           "--warmup", "1", "--configs", "none", "--no-extras", "--no-cpu-baseline", "--inflight", "1"] + list(extra)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env, cwd=ROOT)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stdout[-2000:], p.stderr[-3000:])
    return json.loads(lines[0])


def test_two_ranks_weak_scaling_every_rank_its_own_batch():
    d = _bench("--streams", "512", "--scaling", "weak")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 2
    assert d["config"]["parity_ok"] is True and d["config"]["verified_vs_oracle"] is True
    assert d["config"]["streams_this_rank"] == 512 and d["config"]["streams_whole_job"] == 1024
    assert d["value"] > 0 and d["roofline"]["frac"] > 0


def test_two_ranks_strong_scaling_one_mixed_batch_partitioned_by_the_library():
    """BASELINE.json configs[3] in small: ONE mixed LZ10 / LZ11 / Yaz0 / PRS batch, alz_partition_batch decides which rank decodes what."""
    d = _bench("--streams", "512", "--scaling", "strong", "--format", "mixed")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["parity_ok"] is True
    assert d["config"]["streams_whole_job"] == 512 and 0 < d["config"]["streams_this_rank"] < 512


def test_two_ranks_compress_their_own_buffers():
    """BASELINE.json configs[4] ("LZSS compression ... 1 -> 8 GPUs scaling") as a two-rank job: device-resident encode per rank,
    round trip on the device, the first buffers byte for byte against the oracle's restatement of the managed encoder."""
    d = _bench("--mode", "encode", "--streams", "256", "--quality", "8")
    assert d["n_gpus"] == 2 and d["config"]["mode"] == "encode" and d["config"]["quality"] == 8
    assert d["config"]["parity_ok"] is True and d["config"]["verified_roundtrip_and_vs_oracle"] is True
    assert d["config"]["streams_whole_job"] == 512 and 0.1 < d["config"]["ratio"] < 0.6
