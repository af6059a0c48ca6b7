"""usage (GPU box, library built with -DALZ_CU_DEBUG=1): python tools/cu_debug.py quality -- how often kernel A's chunks had to be cut"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from auroralib.compression_amd import _abi as A, synth, _lib
from auroralib.compression_amd.batch import Context
q = int(sys.argv[1]); n, size = 10000, 262144
ctx = Context(0)
b = synth.make_batch(A.FMT_LZSS, n, size, synth.seed_for(5))
raw, res = ctx.decode_batch(b.streams, b.src, b.dst_bytes)
recs = synth.stream_records(b.streams)
cap = size + size // 4 + 64
streams = (A.Stream * n)()
r2 = synth.stream_records(streams)
r2["src_off"], r2["src_len"] = recs["dst_off"], size
r2["dst_off"] = np.arange(n, dtype=np.uint64) * np.uint64((cap + 255) // 256 * 256)
r2["dst_cap"], r2["format"] = cap, A.FMT_LZSS
lib = ctypes.CDLL(_lib.SO_PATH)
out = (ctypes.c_ulonglong * 8)()
lib.alz_cu_debug_counters(out, 1)
ctx.encode_batch(streams, raw, int(r2["dst_off"][-1]) + cap + 64, quality=q)
lib.alz_cu_debug_counters(out, 1)
print("q%d: chunks %d failed %d | halves %d failed %d | position-by-position chunks %d | narrow %d | kernel B: %.1f of 64 lanes with a candidate over %d trips" % (q, out[0], out[2], out[1], out[3], out[4], out[5], out[6] / max(1, out[7]), out[7]))
