"""ISA filter: every VOP2 `v_cndmask_b32_e32 vD, src0, vS1, vcc` becomes the VOP3 encoding `v_cndmask_b32_e64 vD, src0, vS1, vcc` (same operands, same result).
tools/ubench_issue.hip measured the VOP2 form at 0.175 wave64 instructions per cycle per CU against 0.95 for the VOP3 form (profiles/r05_issue_ceiling.md).
A 32-bit literal as src0 has no VOP3 encoding on gfx950: such lines stay."""
import re
import sys
pat = re.compile(r"^(\s*)v_cndmask_b32_e32 (v\d+), ([^,]+), (v\d+), vcc\s*$")
inline = re.compile(r"^(v\d+|s\d+|-?\d+|0|vcc_lo|vcc_hi|[-+]?(0\.5|1\.0|2\.0|4\.0))$")
n = k = 0
for line in sys.stdin:
    m = pat.match(line.rstrip("\n"))
    if m:
        n += 1
        src0 = m.group(3).strip()
        ok = bool(inline.match(src0))
        if ok and re.match(r"^-?\d+$", src0):
            ok = -16 <= int(src0) <= 64
        if ok:
            k += 1
            line = "%sv_cndmask_b32_e64 %s, %s, %s, vcc\n" % (m.group(1), m.group(2), src0, m.group(4))
    sys.stdout.write(line)
sys.stderr.write("v_cndmask_b32_e32: %d found, %d re-encoded as VOP3\n" % (n, k))
