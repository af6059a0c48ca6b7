"""usage (GPU box): python tools/scan_on_off.py [fmt:quality ...]  -- the encoder's scan path (csrc/alz_encode.hip, enc_scan_select_kernel) on (its probe decides per stream) and off
(alz_debug_scan_mode 0 / 2): kernel time of ONE device-resident encode call over 10 000 windows of 256 KiB of Test.bmp (bench.py's realistic_compress workload) and over the synthetic
cfg5 batch, streams taken, and that both calls leave the same bytes."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from auroralib.compression_amd import _abi as A, synth, formats as F
from auroralib.compression_amd.batch import Context

ctx = Context(0)
ctx.lib.alz_debug_scan_streams.restype = C.c_uint64; ctx.lib.alz_debug_scan_streams.argtypes = [C.c_void_p]
ctx.lib.alz_debug_scan_mode.argtypes = [C.c_void_p, C.c_int]
n, size = int(os.environ.get("N", "10000")), 262144
bmp = np.frombuffer(F.LZSS(A.LzProperties.from_bits(10, 6, 2)).Decompress(open(os.path.join(ROOT, "tests", "golden", "Test.lz"), "rb").read()), dtype=np.uint8)
raw = np.zeros(n * size + 64, dtype=np.uint8)
for i in range(n):
    s0 = (i * (len(bmp) - size)) // max(n - 1, 1)
    raw[i * size:(i + 1) * size] = bmp[s0:s0 + size]
b = synth.make_batch(A.FMT_LZSS, n, size, synth.seed_for(5))
syn, _ = ctx.decode_batch(b.streams, b.src, b.dst_bytes)
syn = np.concatenate([syn[:n * size], np.zeros(64, dtype=np.uint8)])
cap = size + size // 4 + 64
pitch = (cap + 255) // 256 * 256
for spec in (sys.argv[1:] or ["yaz0:8", "lz10:8", "lz11:8", "yay0:8", "lzss:8", "yaz0:5", "lz4_block:8", "snappy_raw:8"]):
    name, q = spec.split(":"); q = int(q); fmt = A.FORMAT_NAMES.index(name)
    for label, data in (("Test.bmp windows", raw), ("synthetic (cfg5)", syn)):
        st = (A.Stream * n)(); r = synth.stream_records(st)
        r["src_off"], r["src_len"] = np.arange(n, dtype=np.uint64) * np.uint64(size), size
        r["dst_off"], r["dst_cap"], r["format"] = np.arange(n, dtype=np.uint64) * np.uint64(pitch), cap, fmt
        d_src, d_dst = ctx.malloc(data.nbytes + 64), ctx.malloc(n * pitch + 64)
        ctx.h2d(d_src, data)
        out = {}
        for mode in (0, 2):
            ctx.lib.alz_debug_scan_mode(ctx.h, mode)
            before = ctx.lib.alz_debug_scan_streams(ctx.h)
            ms = []
            for _ in range(3):
                res, aux = ctx.encode_batch_device(st, d_src, data.nbytes, d_dst, n * pitch + 64, quality=q)
                ms.append(ctx.last_kernel_ms())
            taken = (ctx.lib.alz_debug_scan_streams(ctx.h) - before) // 3
            rr = synth.result_records(res)
            sample = [bytes(ctx.d2h(d_dst, int(rr["dst_len"][i]), offset=i * pitch)) for i in range(0, n, max(1, n // 64))]
            out[mode] = (min(ms), taken, int(rr["dst_len"].astype(np.int64).sum()), bool((rr["status"] == 0).all()), sample)
        ctx.lib.alz_debug_scan_mode(ctx.h, 0)
        ctx.free(d_src); ctx.free(d_dst); ctx.release_scratch()
        same = out[0][2] == out[2][2] and out[0][4] == out[2][4] and out[0][3] and out[2][3]
        print("%-11s q%-2d %-17s scan on %7.1f ms (%5d streams taken)  off %7.1f ms   %.2f x   same bytes %s" % (name, q, label, out[0][0], out[0][1], out[2][0], out[2][0] / out[0][0], same), flush=True)
