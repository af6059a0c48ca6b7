#!/usr/bin/env python3
"""usage: trace_by_grid.py OUT.md TITLE MIN_WORKGROUPS --stats DIR [--fetch DIR] [--write DIR] [--cmd TEXT]
Kernel times and HBM traffic of the dispatches of alz_* kernels that have at least MIN_WORKGROUPS workgroups (a bench.py run that
carries a small headline batch and ONE large named configuration: the large dispatches are the configuration's), from rocprofv3
--kernel-trace and separate --pmc FETCH_SIZE / WRITE_SIZE passes.  FETCH_SIZE x2 (gfx950 rule), WRITE_SIZE x1, per launch."""
import argparse, collections, csv, glob, os


def rows(d, suffix):
    out = []
    for f in sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)):
        out += list(csv.DictReader(open(f, newline="")))
    return out


ap = argparse.ArgumentParser()
ap.add_argument("out"); ap.add_argument("title"); ap.add_argument("minwg", type=int)
ap.add_argument("--stats"); ap.add_argument("--fetch"); ap.add_argument("--write"); ap.add_argument("--cmd", default="")
a = ap.parse_args()
lines = ["# rocprofv3 summary: " + a.title, ""]
if a.cmd:
    lines += ["command: `%s`" % a.cmd, "", "(dispatches of alz_* kernels with at least %d workgroups: the named configuration; the small headline batch of the same run is left out)" % a.minwg, ""]
per = collections.defaultdict(list)
for r in rows(a.stats, "kernel_trace.csv"):
    if "alz_decode" not in r["Kernel_Name"]:
        continue
    wg = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    if wg >= a.minwg:
        per[(r["Kernel_Name"].split("(")[0].replace("void ", ""), wg, r["LDS_Block_Size"], r["VGPR_Count"], r["Scratch_Size"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
lines += ["## kernel times (`rocprofv3 --kernel-trace`)", "", "| kernel | workgroups | calls | avg ms | min ms | max ms | LDS B | VGPR | scratch |", "|---|---|---|---|---|---|---|---|---|"]
for (k, wg, lds, vg, sc), v in sorted(per.items()):
    lines.append("| `%s` | %d | %d | %.4f | %.4f | %.4f | %s | %s | %s |" % (k, wg, len(v), sum(v) / len(v), min(v), max(v), lds, vg, sc))
tot = {}
for label, d, mult in (("FETCH_SIZE", a.fetch, 2.0), ("WRITE_SIZE", a.write, 1.0)):
    if not d:
        continue
    acc = collections.defaultdict(list)
    for r in rows(d, "counter_collection.csv"):
        if "alz_decode" in r["Kernel_Name"] and r["Counter_Name"] == label and int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"])) >= a.minwg:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    tot[label] = {k: sum(v) / len(v) * 1024 * mult for k, v in acc.items()}
    lines += ["", "## %s (separate `--pmc %s` pass)" % (label, label), "", "corrected bytes per launch (x%.0f, gfx950 rule), per kernel:" % mult, ""]
    lines += ["- `%s`: %.0f" % (k.replace("void ", ""), b) for k, b in sorted(tot[label].items())]
if len(tot) == 2:
    lines += ["", "## HBM traffic per launch = 2 x FETCH_SIZE + WRITE_SIZE, per kernel", ""]
    for k in sorted(tot["FETCH_SIZE"]):
        t = int(tot["FETCH_SIZE"][k] + tot["WRITE_SIZE"].get(k, 0))
        lines.append("- `%s`: %d bytes" % (k.replace("void ", ""), t))
        print("TRAFFIC %s %s %d" % (a.title.split()[0], k.replace("void ", ""), t))
open(a.out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[-14:]))
