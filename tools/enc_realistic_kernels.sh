# usage (GPU box): bash tools/enc_realistic_kernels.sh fmt quality [n] [size] -- per-kernel times of ONE device-resident encode call over n (10 000) windows of size (262144) bytes of Test.bmp
# (bench.py's realistic_compress_<fmt>_q<Q> workload; rocprofv3 --kernel-trace --stats; ALZ_SEG=0: without the segmented parse + emit of a small batch)
cd "${GRAFT_REPO_ROOT:?}" || exit 1
export TMPDIR=/tmp
N=${3:-10000}; SZ=${4:-262144}
D=gpurun_out/encr_k_$1_q$2_$N; rm -rf $D; mkdir -p $D
cat > $D/run.py <<PY
import os, sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from auroralib.compression_amd import _abi as A, synth, formats as F
from auroralib.compression_amd.batch import Context
fmt = A.FORMAT_NAMES.index("$1"); n, size = $N, $SZ
ctx = Context(0)
if os.environ.get("ALZ_SEG") is not None: ctx.lib.alz_debug_seg_max_streams(ctx.h, int(os.environ["ALZ_SEG"]))      # (0: the segmented parse + emit off)
lz = F.LZSS(A.LzProperties.from_bits(10, 6, 2))
bmp = np.frombuffer(lz.Decompress(open("tests/golden/Test.lz", "rb").read()), dtype=np.uint8)
if os.environ.get("ALZ_REPEAT"): bmp = np.tile(bmp, int(os.environ["ALZ_REPEAT"]))                     # (the file so many times over: windows larger than it)
starts = [(i * (len(bmp) - size)) // max(n - 1, 1) for i in range(n)]
if os.environ.get("ALZ_STEP"): starts = [(i * int(os.environ["ALZ_STEP"])) % (len(bmp) - size) for i in range(n)]      # (windows ALZ_STEP bytes apart, as tools/mid_batch_encode.py takes them)
raw = np.zeros(n * size + 64, dtype=np.uint8)
for i, s0 in enumerate(starts): raw[i * size:(i + 1) * size] = bmp[s0:s0 + size]
cap = size + size // 4 + 64
st = (A.Stream * n)(); r = synth.stream_records(st)
r["src_off"], r["src_len"] = np.arange(n, dtype=np.uint64) * np.uint64(size), size
r["dst_off"] = np.arange(n, dtype=np.uint64) * np.uint64((cap + 255) // 256 * 256)
r["dst_cap"], r["format"] = cap, fmt
dst_bytes = int(r["dst_off"][-1]) + cap + 64
d_src, d_dst = ctx.malloc(raw.nbytes + 64), ctx.malloc(dst_bytes)
ctx.h2d(d_src, raw)
for _ in range(2):
    res, aux = ctx.encode_batch_device(st, d_src, raw.nbytes, d_dst, dst_bytes, quality=$2)
    print("kernel ms", ctx.last_kernel_ms(), flush=True)
PY
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $D/run.py > $D/log.txt 2>&1
grep "kernel ms" $D/log.txt
python3 - $D <<'PY'
import csv,glob,sys
for fn in glob.glob(sys.argv[1]+'/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        if 'enc_' in r['Name']: print('%-60s calls %s total %.2f ms' % (r['Name'][r['Name'].find('enc_'):][:60], r['Calls'], float(r['TotalDurationNs'])/1e6))
PY
find $D -name "*.csv" -size +1M -delete
