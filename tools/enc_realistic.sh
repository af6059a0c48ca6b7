# usage (GPU box): bash tools/enc_realistic.sh kind quality -- per-kernel times of ONE encode call (LZSS) over 10 000 x 256 KiB of
# kind = bmp (the 193 windows of Test.bmp, repeated) | zeros | noise
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
D=gpurun_out/enc_real_$1_q$2; rm -rf $D; mkdir -p $D
cat > $D/run.py <<PY
import os, sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from auroralib.compression_amd import _abi as A, synth, formats as F
from auroralib.compression_amd.batch import Context
n, size = 10000, 262144
ctx = Context(0)
kind = "$1"
if kind == "bmp":
    lz = F.LZSS(A.LzProperties.from_bits(10, 6, 2))
    bmp = np.frombuffer(lz.Decompress(open(os.path.join("tests", "golden", "Test.lz"), "rb").read()), dtype=np.uint8)
    starts = list(range(0, len(bmp) - size + 1, 4096))
    raw = np.empty(n * size + 64, dtype=np.uint8)
    for i in range(n): raw[i * size:(i + 1) * size] = bmp[starts[i % len(starts)]:starts[i % len(starts)] + size]
elif kind == "zeros":
    raw = np.zeros(n * size + 64, dtype=np.uint8)
else:
    raw = np.random.default_rng(1).integers(0, 256, n * size + 64, dtype=np.uint8)
cap = size + size // 4 + 64
streams = (A.Stream * n)()
r2 = synth.stream_records(streams)
r2["src_off"] = np.arange(n, dtype=np.uint64) * np.uint64(size); r2["src_len"] = size
r2["dst_off"] = np.arange(n, dtype=np.uint64) * np.uint64((cap + 255) // 256 * 256)
r2["dst_cap"], r2["format"] = cap, A.FMT_LZSS
enc, res, aux = ctx.encode_batch(streams, raw, int(r2["dst_off"][-1]) + cap + 64, quality=$2)
rr = synth.result_records(res)
print("ok", bool((rr["status"] == 0).all()), "ratio %.4f" % (rr["dst_len"].sum() / (n * size)))
import hashlib; print("sha", hashlib.sha1(bytes(enc[:int(r2["dst_off"][200]) ])).hexdigest())
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $D/run.py > $D/log.txt 2>&1
grep -E "^ok|^sha" $D/log.txt
python3 - $D <<'PY'
import csv,glob,sys
for fn in glob.glob(sys.argv[1]+'/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        if 'enc_' in r['Name']: print('%-40s calls %s total %.1f ms' % (r['Name'][r['Name'].find('enc_'):][:40], r['Calls'], float(r['TotalDurationNs'])/1e6))
PY
find $D -name "*.csv" -size +1M -delete
