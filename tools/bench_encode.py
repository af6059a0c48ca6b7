#!/usr/bin/env python3
"""cfg5: LZSS(12,4,2) compression of 10 000 x 256 KiB raw buffers (obtained by decoding synthetic LZSS streams on the
GPU), host-buffer API (upload + kernels + download).  Kernel-only times come from rocprofv3 --kernel-trace --stats."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from auroralib.compression_amd import _abi as A  # noqa: E402
from auroralib.compression_amd import synth  # noqa: E402
from auroralib.compression_amd.batch import Context  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--format", default="lzss")
ap.add_argument("--streams", type=int, default=10000)
ap.add_argument("--stream-kib", type=int, default=256)
ap.add_argument("--quality", type=int, default=0)
ap.add_argument("--reps", type=int, default=2)
a = ap.parse_args()
fmt = A.FORMAT_NAMES.index(a.format)
n, size = a.streams, a.stream_kib * 1024
ctx = Context(0)
b = synth.make_batch(A.FMT_LZSS, n, size, synth.seed_for(5))
raw, res = ctx.decode_batch(b.streams, b.src, b.dst_bytes)
recs = synth.stream_records(b.streams)
cap = size + size // 4 + 64
streams = (A.Stream * n)()
r2 = synth.stream_records(streams)
r2["src_off"], r2["src_len"] = recs["dst_off"], size
r2["dst_off"] = np.arange(n, dtype=np.uint64) * np.uint64((cap + 255) // 256 * 256)
r2["dst_cap"], r2["format"] = cap, fmt
dst_bytes = int(r2["dst_off"][-1]) + cap + 64
best = None
for _ in range(a.reps):
    t = time.perf_counter()
    dst, eres, aux = ctx.encode_batch(streams, raw, dst_bytes, quality=a.quality)
    dt = time.perf_counter() - t
    best = dt if best is None else min(best, dt)
er = synth.result_records(eres)
ok = bool((er["status"] == 0).all())
comp = int(er["dst_len"].astype(np.int64).sum())
# round trip on the GPU
s3 = (A.Stream * n)()
r3 = synth.stream_records(s3)
r3["src_off"], r3["src_len"], r3["dst_off"], r3["dst_cap"], r3["decom_len"], r3["format"] = r2["dst_off"], er["dst_len"], recs["dst_off"], size, size, fmt
r3["aux0"], r3["aux1"] = np.frombuffer(aux, dtype=np.uint32).reshape(n, 2)[:, 0], np.frombuffer(aux, dtype=np.uint32).reshape(n, 2)[:, 1]
back, dres = ctx.decode_batch(s3, dst, b.dst_bytes)
rt = bool(np.array_equal(back[:b.dst_bytes], raw[:b.dst_bytes]))
print(json.dumps({"workload": "%s compress q%d, %d x %d KiB (decoded synthetic LZSS)" % (a.format, a.quality, n, a.stream_kib),
                  "raw_GiB_per_s_host_api": round(n * size / best / 2**30, 3), "seconds": round(best, 4), "ratio": round(comp / (n * size), 4),
                  "all_ok": ok, "roundtrip_ok": rt}))
