#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (kernel-trace/stats run + separate FETCH_SIZE / WRITE_SIZE pmc runs) into one small
markdown file for profiles/.  Usage: profile_summary.py OUT.md --stats DIR [--fetch DIR] [--write DIR] [--note TEXT]

HBM traffic follows MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads
exactly half of the bytes of a wide coalesced streaming read, so it is doubled; WRITE_SIZE is exact for 16 B/lane stores.
"""
import argparse
import csv
import glob
import os
import sys


def rows(d, suffix):
    out = []
    for f in sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)):
        with open(f, newline="") as fh:
            out += list(csv.DictReader(fh))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--stats")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--pmc", nargs="*", default=[], help="dirs of extra --pmc passes (SQ_*/GRBM_* counters)")
    ap.add_argument("--note", default="")
    ap.add_argument("--cmd", default="")
    a = ap.parse_args()
    lines = ["# rocprofv3 summary", ""]
    if a.note:
        lines += [a.note, ""]
    if a.cmd:
        lines += ["command: `%s`" % a.cmd, ""]
    if a.stats:
        lines += ["## kernel stats (`rocprofv3 --kernel-trace --stats`)", "", "| kernel | calls | avg ms | min ms | max ms | % |", "|---|---|---|---|---|---|"]
        for r in rows(a.stats, "kernel_stats.csv"):
            name = r["Name"].replace("(anonymous namespace)::", "").split("(")[0]
            lines.append("| `%s` | %s | %.4f | %.4f | %.4f | %s |" % (name, r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6, r["Percentage"]))
        tr = [r for r in rows(a.stats, "kernel_trace.csv") if "alz_" in r.get("Kernel_Name", "")]
        if tr:
            r = tr[-1]
            lines += ["", "last dispatch of the dominant kernel: grid %s, workgroup %s, LDS %s B, VGPR %s, SGPR %s, scratch %s" % (
                r.get("Grid_Size_X"), r.get("Workgroup_Size_X"), r.get("LDS_Block_Size"), r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("Scratch_Size"))]
    for label, d, mult in (("FETCH_SIZE", a.fetch, 2.0), ("WRITE_SIZE", a.write, 1.0)):
        if not d:
            continue
        rs = [r for r in rows(d, "counter_collection.csv") if "alz_" in r["Kernel_Name"] and r["Counter_Name"] == label]
        if not rs:
            continue
        # a launch may be several kernels (a work-queue launch has its gated repair kernel behind it, a mixed batch one kernel per format): per kernel the mean over
        # its dispatches, per launch their sum
        import collections
        per = collections.OrderedDict()
        for r in rs:
            per.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
        avg = sum(sum(v) / len(v) for v in per.values())
        lines += ["", "## %s (separate `--pmc %s` pass)" % (label, label), "",
                  "dispatches of alz_* kernels: %d (%d kernels per launch); mean raw counter per launch = %.1f KiB; corrected bytes per launch = %.0f (x%.0f, gfx950 rule)" % (len(rs), len(per), avg, avg * 1024 * mult, mult)]
    if a.pmc:
        import collections
        acc = collections.OrderedDict()
        perk = collections.OrderedDict()
        for d in a.pmc:
            for r in rows(d, "counter_collection.csv"):
                if "alz_" in r["Kernel_Name"]:
                    perk.setdefault(r["Counter_Name"], collections.OrderedDict()).setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
        for k, kk in perk.items():
            acc[k] = [sum(sum(v) / len(v) for v in kk.values())]          # per launch: the kernels' means added up
        lines += ["", "## PMC counters (separate `--pmc` passes, mean per LAUNCH: the alz_* kernels of a launch added up)", "", "| counter | mean |", "|---|---|"]
        for k, v in acc.items():
            lines.append("| %s | %.4g |" % (k, sum(v) / len(v)))
        g = acc.get("GRBM_GUI_ACTIVE"); va = acc.get("SQ_ACTIVE_INST_VALU"); vi = acc.get("SQ_INSTS_VALU")
        if g and va:
            cyc = sum(g) / len(g) / 8.0            # counter is summed over the 8 XCDs
            busy = sum(va) / len(va) * 4.0 / 1024.0 / cyc   # SQ_ACTIVE_* count quad-cycles; 1024 SIMDs
            lines += ["", "derived: %.3g shader cycles per dispatch; VALU busy = SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / cycles = %.0f %%%s" % (cyc, busy * 100, " (an UPPER bound: the counter is 1 per instruction, and add / sub / and / or / xor / shift-right / mov occupy the SIMD for ~2.2 cycles, not 4 -- profiles/r05_issue_ceiling.md; bench.py's `roofline.issue` prices all three pipes against the measured ceilings)")]
    open(a.out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
