"""usage: python tools/update_encode_counters.py gpurun_out/traffic_encode.txt  -- takes the three `"lzss_encode_q<Q>:10000:256": <bytes>` lines tools/traffic_encode.sh printed (cfg5 at
quality 0 / 8 / 15: FETCH_SIZE x 2 + WRITE_SIZE of the compression kernels, separate --pmc passes) into profiles/traffic.json, keeps the log under profiles/, and stamps both counter files
with the hash of the encoder sources they were measured on (tools/kernel_hash.py, family "encode").  Run it on the checkout the counters came from:
    gpurun -- 'bash tools/traffic_encode.sh > gpurun_out/traffic_encode.txt 2>&1'; python tools/update_encode_counters.py gpurun_out/traffic_encode.txt"""
import json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_hash

tp = os.path.join(ROOT, "profiles", "traffic.json")
traffic = json.load(open(tp))
found = dict(re.findall(r'"(lzss_encode_q\d+:10000:256)": (\d+)', open(sys.argv[1]).read()))
if sorted(found) != ["lzss_encode_q0:10000:256", "lzss_encode_q15:10000:256", "lzss_encode_q8:10000:256"]:
    sys.exit("not the three lines of tools/traffic_encode.sh: %r" % sorted(found))
for k, v in found.items():
    traffic[k] = int(v)
    print(k, v)
json.dump(traffic, open(tp, "w"), indent=1)
shutil.copy(sys.argv[1], os.path.join(ROOT, "profiles", "r06_traffic_encode.txt"))
kernel_hash.stamp(("encode",))
print(json.dumps(kernel_hash.current()))
