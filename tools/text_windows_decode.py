"""Decode throughput on program text and prose: 4 096 windows of 256 KiB of the repository's own sources (1.7 MB, so the windows overlap), encoded
on the GPU at quality 8, decoded as one device-resident batch -- beside bench.py's realistic_* entries (a bitmap) and the synthetic bodies.
(LZO reports "round trip False": some windows start with a short run, where the reference's encoder writes a stream its decoder does not read
back -- tests/test_oracle_golden.py::test_lzo_two_literal_runs_in_a_row_is_the_reference; the GPU's bytes are the reference's.)"""
import os, sys, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from auroralib.compression_amd import _abi as A
from auroralib.compression_amd import synth
from auroralib.compression_amd.batch import Context, Plan
files = sorted(f for pat in ("*.md", "*.hip", "*.h", "*.py", "*.cs", "*.cpp") for f in glob.glob(os.path.join(ROOT, "**", pat), recursive=True) if "gpurun_out" not in f)
text = np.frombuffer(b"".join(open(f, "rb").read() for f in files), dtype=np.uint8)
n, size = 4096, 262144
starts = [(i * (len(text) - size)) // (n - 1) for i in range(n)]
raw = np.concatenate([text[s:s + size] for s in starts])
c = Context(0)
for fname in sys.argv[1:] or ["yaz0", "lz10", "lz11", "prs_be", "lz4_block", "lzo", "snappy_raw"]:
    fmt = A.FORMAT_NAMES.index(fname)
    cap = size + size // 4 + 64
    st = (A.Stream * n)()
    r = synth.stream_records(st)
    r["src_off"], r["src_len"] = np.arange(n, dtype=np.uint64) * np.uint64(size), size
    r["dst_off"] = np.arange(n, dtype=np.uint64) * np.uint64((cap + 255) // 256 * 256)
    r["dst_cap"], r["format"] = cap, fmt
    dst_bytes = int(r["dst_off"][-1]) + cap + 64
    d_raw, d_comp, d_out = c.malloc(raw.nbytes + 64), c.malloc(dst_bytes), c.malloc(raw.nbytes + 64)
    try:
        c.h2d(d_raw, raw)
        res, aux = c.encode_batch_device(st, d_raw, raw.nbytes, d_comp, dst_bytes, quality=8)
        rr = synth.result_records(res)
        assert (rr["status"] == 0).all()
        sized = fname not in ("lz4_block", "prs_be", "prs_le", "lzo", "snappy_raw")
        ds = (A.Stream * n)()
        d = synth.stream_records(ds)
        d["src_off"], d["src_len"] = r["dst_off"], rr["dst_len"]
        d["dst_off"] = np.arange(n, dtype=np.uint64) * np.uint64(size)
        d["dst_cap"], d["decom_len"], d["format"] = size, size if sized else 0, fmt
        auxv = np.frombuffer(aux, dtype=np.uint32).reshape(n, 2)
        d["aux0"], d["aux1"] = auxv[:, 0], auxv[:, 1]
        p = Plan(c, ds)
        p.execute(d_comp, d_out); c.synchronize()
        ms = p.execute_timed(d_comp, d_out, iters=10)
        dres = synth.result_records(p.results())
        ok = bool((dres["status"] == 0).all()) and bool(np.array_equal(c.d2h(d_out, raw.nbytes), raw))
        p.close()
    finally:
        c.free(d_raw); c.free(d_comp); c.free(d_out)
    comp = int(rr["dst_len"].astype(np.int64).sum())
    print("%-10s %d text windows x 256 KiB (ratio %.3f): decode %.3f ms = %.0f GiB/s, round trip %s" % (fname, n, comp / (n * size), ms, n * size / ms / 2**30 * 1e3, ok), flush=True)
