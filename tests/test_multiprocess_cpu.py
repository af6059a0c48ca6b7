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


STRONG = textwrap.dedent('''
    import os, sys, json
    sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import numpy as np, torch, torch.distributed as dist
    import oracle_lib as O
    from auroralib.compression_amd import _abi as A, synth
    from auroralib.compression_amd.batch import partition_batch
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    # strong scaling (bench.py --scaling strong): ONE mixed batch, split by the library's LPT partitioner; a rank generates
    # and decodes exactly its share, seeded by the GLOBAL stream index
    ntot, target = 64, 6000
    fm = np.array([[A.FMT_LZ10, A.FMT_LZ11, A.FMT_YAZ0, A.FMT_PRS_BE][i %% 4] for i in range(ntot)], dtype=np.uint32)
    table = (A.Stream * ntot)()
    tr = synth.stream_records(table)
    tr["dst_cap"], tr["decom_len"], tr["format"] = target, target, fm
    part, cost = partition_batch(table, world)
    mine = np.nonzero(part == rank)[0]
    b = synth.make_batch(fm[mine], len(mine), target, 0, seeds=(synth.seed_for(4) + mine).astype(np.uint64))
    dst, res = O.decode_batch(b.streams, b.src, b.dst_bytes, nthreads=1)
    rr, sr = synth.result_records(res), synth.stream_records(b.streams)
    assert (rr["status"] == 0).all() and (rr["dst_len"] == target).all()
    h = torch.zeros(ntot, dtype=torch.float64)
    for j, g in enumerate(mine):
        a = int(sr["dst_off"][j]); h[g] = float(O.xxh64(dst[a:a + target].tobytes()) %% (1 << 50))
    dist.all_reduce(h)                                       # every stream was decoded by exactly one rank
    if rank == 0:
        whole = synth.make_batch(fm, ntot, target, synth.seed_for(4))
        wd, wr = O.decode_batch(whole.streams, whole.src, whole.dst_bytes, nthreads=2)
        ws = synth.stream_records(whole.streams)
        ref = [float(O.xxh64(wd[int(ws["dst_off"][i]):int(ws["dst_off"][i]) + target].tobytes()) %% (1 << 50)) for i in range(ntot)]
        assert ref == h.tolist()
        print(json.dumps({"world": world, "shares": np.bincount(part, minlength=world).tolist(), "imbalance": float((cost.max() - cost.min()) / cost.mean())}))
    dist.barrier(); dist.destroy_process_group()
''')


def test_strong_scaling_share_of_one_batch_gloo(tmp_path):
    script = tmp_path / "strong.py"
    script.write_text(STRONG % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["world"] == 2 and sum(d["shares"]) == 64 and min(d["shares"]) > 20 and d["imbalance"] < 0.05


def test_bench_refuses_a_gpu_count_that_is_not_the_world_size():
    """--gpus used to be parsed and ignored.  Under a launcher whose WORLD_SIZE differs it now fails loudly
    (before touching any GPU); without a launcher it starts the one-process-per-GPU job itself (needs GPUs: -m gpu)."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 2 and "WORLD_SIZE" in out.stderr
