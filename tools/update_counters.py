"""usage: python tools/update_counters.py gpurun_out/r06_<name>.md ...  -- takes the per-launch HBM bytes (FETCH_SIZE x 2 + WRITE_SIZE) and the instruction counts (SQ_INSTS_VALU /
SALU / LDS) out of the summaries tools/gpu_profile.sh wrote, puts them into profiles/traffic.json / profiles/insts.json under the workload's key, copies the summaries to profiles/,
and stamps both files with the hash of the kernel sources they were measured on (tools/kernel_hash.py).  Run it on the checkout the counters came from.
A name is r06_<format> (10 000 x 256 KiB of that format), r06_cfg2 (yaz0:10000:64), r06_cfg4_shard (mixed:5000:256) or r06_cfg3_100000 (lz4_block:100000:256)."""
import json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_hash

KEYS = {"cfg2": "yaz0:10000:64", "cfg4_shard": "mixed:5000:256", "cfg3_100000": "lz4_block:100000:256"}
tp, ip = os.path.join(ROOT, "profiles", "traffic.json"), os.path.join(ROOT, "profiles", "insts.json")
traffic, insts = json.load(open(tp)), json.load(open(ip))
for path in sys.argv[1:]:
    name = os.path.basename(path)[:-3]
    m = re.match(r"r\d+_(.*)", name)
    key = KEYS.get(m.group(1), "%s:10000:256" % m.group(1))
    text = open(path).read()
    b = [float(x) for x in re.findall(r"corrected bytes per launch = (\d+)", text)]
    if len(b) == 2:
        traffic[key] = int(b[0] + b[1])
    c = {k: float(v) for k, v in re.findall(r"\| (SQ_INSTS_VALU|SQ_INSTS_SALU|SQ_INSTS_LDS) \| ([0-9.e+]+) \|", text)}
    if len(c) == 3:
        insts[key] = {"valu": c["SQ_INSTS_VALU"], "salu": c["SQ_INSTS_SALU"], "lds": c["SQ_INSTS_LDS"], "source": "profiles/%s.md" % name}
    shutil.copy(path, os.path.join(ROOT, "profiles", name + ".md"))
    print(key, traffic.get(key), insts.get(key))
json.dump(traffic, open(tp, "w"), indent=1); open(tp, "a").write("\n")
json.dump(insts, open(ip, "w"), indent=1); open(ip, "a").write("\n")
kernel_hash.stamp(("decode",))
print(json.dumps(kernel_hash.current()))
