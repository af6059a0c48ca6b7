# usage (GPU box): bash tools/enc_kernels.sh fmt quality -- per-kernel times of one encode call (rocprofv3 --kernel-trace --stats)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
D=gpurun_out/enc_k_$1_q$2; rm -rf $D; mkdir -p $D
cat > $D/run.py <<PY
import sys, numpy as np
sys.path.insert(0, '.')
from auroralib.compression_amd import _abi as A, synth
from auroralib.compression_amd.batch import Context
fmt = A.FORMAT_NAMES.index("$1"); n, size = 10000, 262144
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
ctx.encode_batch(streams, raw, int(r2["dst_off"][-1]) + cap + 64, quality=$2)
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $D/run.py > $D/log.txt 2>&1
python3 - $D <<'PY'
import csv,glob,sys
for fn in glob.glob(sys.argv[1]+'/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        if 'enc_' in r['Name']: print('%-40s calls %s total %.1f ms' % (r['Name'][r['Name'].find('enc_'):][:40], r['Calls'], float(r['TotalDurationNs'])/1e6))
PY
find $D -name "*.csv" -size +1M -delete
