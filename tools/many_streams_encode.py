"""usage (GPU box): python tools/many_streams_encode.py -- 70 000 streams in one alz_encode_batch call (more than one gridDim.y can hold): statuses, and samples around the 65 535th against the CPU restatement"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from auroralib.compression_amd import _abi as A, synth
from gpu_common import ctx
n, size = 70000, 2048
rng = np.random.default_rng(3)
raw = rng.integers(0, 8, n * size + 64, dtype=np.uint8)
streams = (A.Stream * n)()
r = synth.stream_records(streams)
cap = size + size // 4 + 64
r["src_off"] = np.arange(n, dtype=np.uint64) * np.uint64(size); r["src_len"] = size
r["dst_off"] = np.arange(n, dtype=np.uint64) * np.uint64((cap + 15) // 16 * 16); r["dst_cap"], r["format"] = cap, A.FMT_LZSS
for q in (0, 8):
    dst, res, aux = ctx().encode_batch(streams, raw, int(r["dst_off"][-1]) + cap + 64, quality=q)
    rr = synth.result_records(res)
    bad = 0
    for i in (0, 1, 65534, 65535, 65536, 69999):
        want, _ = O.encode_stream(A.FMT_LZSS, bytes(raw[i * size:(i + 1) * size]), quality=q)
        got = bytes(dst[int(r["dst_off"][i]):int(r["dst_off"][i]) + int(rr["dst_len"][i])])
        bad += got != want
    print("q%d: %d streams, all ok %s, sample mismatches %d" % (q, n, bool((rr["status"] == 0).all()), bad))
