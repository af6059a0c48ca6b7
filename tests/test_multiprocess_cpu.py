"""CPU, world_size 2, gloo: the multi-GPU path of bench.py shards streams with no data-path collective -- the only
distributed operations are the barrier and the MAX-reduce of the step time.  This test runs the same sharding /
reduction logic with two processes on the CPU (the decode itself is replaced by the oracle here because no GPU exists in
this container; the GPU path is covered by -m gpu tests)."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys, json
    sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import numpy as np, torch, torch.distributed as dist
    import oracle_lib as O
    from auroralib.compression_amd import _abi as A, synth
    from auroralib.compression_amd.sharding import shard_seed, reduce_step_time, whole_job_value
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n, target = 24, 8192
    b = synth.make_batch(A.FMT_YAZ0, n, target, shard_seed(2, rank, n))
    dst, res = O.decode_batch(b.streams, b.src, b.dst_bytes, nthreads=1)
    rr = synth.result_records(res)
    assert (rr["status"] == 0).all() and (rr["dst_len"] == target).all()
    # ranks decode DISJOINT streams (different seeds -> different bytes)
    digest = torch.tensor([float(O.xxh64(dst.tobytes()) %% (1 << 40))], dtype=torch.float64)
    gathered = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(gathered, digest)
    assert len({float(g.item()) for g in gathered}) == world
    dt = reduce_step_time(0.010 * (rank + 1), dist)          # MAX over ranks
    assert abs(dt - 0.010 * world) < 1e-9
    v = whole_job_value(n * target, world, steps=1, seconds=dt)
    if rank == 0:
        print(json.dumps({"world": world, "value": v, "dt": dt}))
    dist.barrier(); dist.destroy_process_group()
''')


def test_two_process_sharding_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29531", str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    import json
    d = json.loads(line)
    assert d["world"] == 2 and abs(d["value"] - 24 * 8192 * 2 / 0.020 / 2**30) < 1e-9
