"""usage: python3 tools/isa_blocks.py KERNEL_SUBSTRING [source.hip]  -- a per-source-function instruction table of ONE kernel from its ISA.
Compiles the source for gfx950 with line tables (-gline-tables-only changes no code), attributes every instruction of the kernel to the source function its `.loc` lies in
(inlined code keeps its own lines) and counts vector / scalar / LDS / memory / other instructions, in all and inside loops (`Depth=` of the block).  Static counts: what the
binary holds, not what a launch executes -- the launch's totals are in profiles/r05_<format>.md (PMC)."""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = os.path.join(ROOT, "auroralib", "compression_amd", "csrc")
want = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else "alz_kernels.hip"
out = "/tmp/isa_blocks_%s.s" % os.path.basename(src)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-gline-tables-only", "-fPIC", "--offload-arch=gfx950", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + C,
                       "-Wno-unused-function", "-Wno-inline-asm", "-x", "hip", "--cuda-device-only", "-S", os.path.join(C, src), "-o", out], stderr=subprocess.DEVNULL)
L = open(out).read().split("\n")
files = {}
for l in L:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', l)
    if m: files[int(m.group(1))] = os.path.join(m.group(2), m.group(3)) if not os.path.isabs(m.group(3)) else m.group(3)
def functions_of(path):
    """line -> name of the function (or struct method) whose definition starts at or before it"""
    full = path if os.path.isabs(path) else os.path.join(ROOT, path)
    starts = []
    try: text = open(full).read().split("\n")
    except OSError: return lambda ln: os.path.basename(path)
    for i, t in enumerate(text, 1):
        if re.match(r'\s*(?:template\s*<[^>]*>\s*)?(?:static\s+|inline\s+|__device__\s+|__global__\s+|__forceinline__\s+|__launch_bounds__\(\d+\)\s+)+[\w:<>\*&\s,]*?\b([A-Za-z_]\w*)\s*\(', t) and not t.strip().startswith(("return", "if", "for", "while")):
            name = re.match(r'.*?\b([A-Za-z_]\w*)\s*\(', re.sub(r'__launch_bounds__\(\d+\)', '', t)).group(1)
            starts.append((i, name))
    def f(ln):
        name = os.path.basename(path)
        for i, nme in starts:
            if i <= ln: name = nme
            else: break
        return name
    return f
fn_cache = {}
start = next(i for i, l in enumerate(L) if want in l and l.rstrip().endswith(":") is False and re.match(r'^_Z\w+:', l) and want in l)
end = next(i for i in range(start, len(L)) if "s_endpgm" in L[i])
print("kernel: %s  (%d lines of assembly)" % (L[start].split(":")[0], end - start))
tab = collections.defaultdict(lambda: collections.Counter())
cur_fn, depth = "?", 0
for i in range(start + 1, end + 1):
    t = L[i].strip()
    m = re.match(r'\.loc\s+(\d+)\s+(\d+)', t)
    if m:
        fno, ln = int(m.group(1)), int(m.group(2))
        path = files.get(fno, "?")
        if path not in fn_cache: fn_cache[path] = functions_of(path)
        cur_fn = "%s: %s" % (os.path.basename(path), fn_cache[path](ln))
        continue
    m = re.match(r'\.LBB\d+_\d+:(.*)', t)
    if m:
        d = re.search(r'Depth=(\d+)', m.group(1)); depth = int(d.group(1)) if d else 0
        continue
    if not t or t.startswith((";", ".")): continue
    op = t.split()[0]
    k = "salu" if op.startswith("s_") else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else "mem" if op.startswith(("global", "flat", "buffer", "scratch")) else "other"
    tab[cur_fn][k] += 1
    tab[cur_fn]["all"] += 1
    if depth: tab[cur_fn]["loop"] += 1
tot = collections.Counter()
print("| source function | instructions | vector | scalar | LDS | memory | inside loops |\n|---|---|---|---|---|---|---|")
for fn, c in sorted(tab.items(), key=lambda kv: -kv[1]["all"]):
    if c["all"] < 8: tot.update({"rest_" + k: v for k, v in c.items()}); continue
    print("| `%s` | %d | %d | %d | %d | %d | %d |" % (fn, c["all"], c["valu"], c["salu"], c["lds"], c["mem"], c["loop"]))
    tot.update(c)
print("| (functions with fewer than 8) | %d | %d | %d | %d | %d | %d |" % (tot["rest_all"], tot["rest_valu"], tot["rest_salu"], tot["rest_lds"], tot["rest_mem"], tot["rest_loop"]))
print("| all | %d | %d | %d | %d | %d | %d |" % (tot["all"] + tot["rest_all"], tot["valu"] + tot["rest_valu"], tot["salu"] + tot["rest_salu"], tot["lds"] + tot["rest_lds"], tot["mem"] + tot["rest_mem"], tot["loop"] + tot["rest_loop"]))
