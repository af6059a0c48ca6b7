# usage (GPU box): bash tools/emit_stats.sh fmt...  -- experiment build (-DALZ_EMIT_STATS): per-step averages of the byte phase
# (steps per stream, passes per step, chunks per step); rebuilds the product library afterwards
cd $GRAFT_REPO_ROOT
rm -rf auroralib/compression_amd/csrc/_obj
ALZ_EXTRA_FLAGS="-DALZ_EMIT_STATS" bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
for f in ${@:-yaz0}; do
python3 - $f <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
from auroralib.compression_amd import _abi as A, synth
from auroralib.compression_amd.batch import Context
f = sys.argv[1]
fmt = A.FORMAT_NAMES.index(f)
b = synth.make_batch(fmt, 512, 262144, synth.seed_for(2))
with Context(0) as ctx:
    g_dst, g_res = ctx.decode_batch(b.streams, b.src, b.dst_bytes)
r = synth.result_records(g_res)["reserved"].astype(np.int64)
steps = r & 0xFFFF; passes = ((r >> 16) & 0xFF) / 16.0; chunks = (r >> 24) / 2.0
print("%-12s steps/stream %.0f  passes/step %.2f  chunks/step %.1f  bytes/step %.0f" % (f, steps.mean(), passes.mean(), chunks.mean(), 262144 / max(1, steps.mean())))
PY
done
rm -rf auroralib/compression_amd/csrc/_obj
bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
