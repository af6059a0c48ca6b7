#!/usr/bin/env python3
"""export_batch.py -- writes the synthetic batch bench.py measures to an ALZB file for the managed baseline harness
(baseline/Program.cs): the managed library then decodes byte-identical inputs.  Host code only (no GPU)."""
import argparse
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from auroralib.compression_amd import _abi as A  # noqa: E402
from auroralib.compression_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--format", default="yaz0")
ap.add_argument("--streams", type=int, default=10000)
ap.add_argument("--stream-kib", type=int, default=256)
ap.add_argument("--config", type=int, default=2, help="seed = 0xA17A0000 + 1000 * config + stream index (SURVEY 8d)")
ap.add_argument("--out", required=True)
a = ap.parse_args()
mixed = ["lz10", "lz11", "yaz0", "prs_be"]
fm = np.array([A.FORMAT_NAMES.index(mixed[i % 4] if a.format == "mixed" else a.format) for i in range(a.streams)], dtype=np.uint32)
b = synth.make_batch(fm, a.streams, a.stream_kib * 1024, synth.seed_for(a.config))
r = synth.stream_records(b.streams)
with open(a.out, "wb") as f:
    f.write(b"ALZB" + struct.pack("<II", 1, a.streams))
    for i in range(a.streams):
        sized = int(r["format"][i]) not in (A.FMT_PRS_BE, A.FMT_PRS_LE, A.FMT_LZ4_BLOCK, A.FMT_LZO, A.FMT_SNAPPY_RAW)
        f.write(struct.pack("<IIIII", int(r["format"][i]), int(r["decom_len"][i]) if sized else 0, int(r["aux0"][i]), int(r["aux1"][i]), int(r["src_len"][i])))
    for i in range(a.streams):
        o = int(r["src_off"][i])
        f.write(b.src[o:o + int(r["src_len"][i])].tobytes())
print("wrote %s: %d streams, %d compressed bytes" % (a.out, a.streams, int(r["src_len"].astype(np.int64).sum())))
