"""Per-kernel totals of a rocprofv3 --kernel-trace CSV, grouped by call (a call starts at the kernel named in argv[2], default benc_setup)."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2] if len(sys.argv) > 2 else "benc_setup"
calls = []
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if n.startswith("__amd"): continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if n.startswith(first): calls.append({})
    if calls: calls[-1][n] = calls[-1].get(n, 0.0) + d
for i, c in enumerate(calls):
    if i % 4 == 3:
        print(i // 4, "total %.0f us |" % sum(c.values()), " | ".join("%s %.0f" % (k[:28], v) for k, v in c.items()))
