"""usage (GPU box, with build/variants/phase_times.so built from tools/variants/r04_phase_times.patch):
    python tools/phase_times.py [fmt ...]
Where a wavefront of the flag-family kernel spends its time: s_memtime samples around the front end (input window -> 64 tokens), the token
prologue and the byte phase of every iteration, summed per stream, for a lone stream, one resident round (6 144 streams) and the bench batch
(10 000).  The variant library returns the three sums in place of the result fields."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from auroralib.compression_amd import _lib
_lib.SO_PATH = os.path.join(ROOT, "build", "variants", "phase_times.so")
from auroralib.compression_amd import _abi as A, synth
from auroralib.compression_amd.batch import Context, Plan

ctx = Context(0)
ctx.set_kernel_variant(1)      # one wavefront per stream at every batch size (the two-wavefront shape is not instrumented)
for f in (sys.argv[1:] or ["yaz0", "lz10"]):
    fmt = A.FORMAT_NAMES.index(f)
    for n in (1, 1024, 3584, 6144, 10000):
        b = synth.make_batch(fmt, n, 262144, synth.seed_for(2))
        if os.environ.get("ALZ_PHASE_DATA") == "text":         # 16 windows of program text (the repository's own sources), encoded by the oracle at quality 8, repeated
            import glob
            import oracle_lib as O
            files = sorted(f for pat in ("*.md", "*.hip", "*.h", "*.py", "*.cs", "*.cpp") for f in glob.glob(os.path.join(ROOT, "**", pat), recursive=True) if "gpurun_out" not in f and "build/" not in f)
            text = b"".join(open(f, "rb").read() for f in files)
            comps = [O.encode_stream(fmt, text[k * 90000:k * 90000 + 262144], quality=8) for k in range(16)]
            offs, chunks, o = [], [], 0
            for c_, _a in comps:
                offs.append(o); chunks.append(c_ + bytes((-len(c_)) % 16)); o += len(chunks[-1])
            src = np.frombuffer(b"".join(chunks) + bytes(64), dtype=np.uint8).copy()
            rec = synth.stream_records(b.streams)
            w = np.arange(n) % 16
            rec["src_off"] = np.array(offs, dtype=np.uint64)[w]; rec["src_len"] = np.array([len(c_) for c_, _ in comps], dtype=np.uint32)[w]
            rec["aux0"] = np.array([a.aux0 for _, a in comps], dtype=np.uint32)[w]; rec["aux1"] = np.array([a.aux1 for _, a in comps], dtype=np.uint32)[w]
            b.src = src
        d_src, d_dst = ctx.malloc(b.src.nbytes + 64), ctx.malloc(b.dst_bytes + 64)
        ctx.h2d(d_src, b.src)
        p = Plan(ctx, b.streams)
        p.execute(d_src, d_dst); ctx.synchronize()
        ms = p.execute_timed(d_src, d_dst, iters=3)
        r = synth.result_records(p.results())
        p.close(); ctx.free(d_src); ctx.free(d_dst)
        fr, pr, st = r["dst_len"].astype(np.float64), r["src_used"].astype(np.float64), r["reserved"].astype(np.float64)
        tot = fr + pr + st
        print("%-6s %6d streams: kernel %.3f ms | ticks per stream: front end %.3g (%.0f%%)  prologue %.3g (%.0f%%)  byte phase %.3g (%.0f%%)  sum %.3g  [slowest stream %.3g]" % (
            f, n, ms, fr.mean(), 100 * fr.sum() / tot.sum(), pr.mean(), 100 * pr.sum() / tot.sum(), st.mean(), 100 * st.sum() / tot.sum(), tot.mean(), tot.max()), flush=True)
