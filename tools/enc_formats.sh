# usage (GPU box): bash tools/enc_formats.sh [quality] -- encode 10 000 x 256 KiB per format: kernel ms (HIP events inside alz_encode_batch) and host-API rate
cd $GRAFT_REPO_ROOT
q=${1:-8}
for f in ${FORMATS:-lzss lz10 lz11 yaz0 yay0 mio0 prs_be lz4_block lzo snappy_raw}; do
python3 - $f $q <<'PY'
import sys, time, numpy as np
sys.path.insert(0, '.')
from auroralib.compression_amd import _abi as A, synth
from auroralib.compression_amd.batch import Context
f, q = sys.argv[1], int(sys.argv[2])
fmt = A.FORMAT_NAMES.index(f)
n, size = 10000, 262144
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
ctx.encode_batch(streams, raw, dst_bytes, quality=q)
t = time.perf_counter()
dst, eres, aux = ctx.encode_batch(streams, raw, dst_bytes, quality=q)
dt = time.perf_counter() - t
er = synth.result_records(eres)
print("%-11s q%d kernels %7.1f ms = %5.1f GiB/s  host api %5.2f GiB/s  ratio %.3f ok %s" % (f, q, ctx.last_kernel_ms(), n * size / ctx.last_kernel_ms() / 2**30 * 1e3, n * size / dt / 2**30, er["dst_len"].sum() / (n * size), bool((er["status"] == 0).all())), flush=True)
PY
done
