"""usage: python tools/kernel_hash.py [--stamp]  -- the hash that ties the committed counter files (profiles/traffic.json, profiles/insts.json) to the kernel sources
they were measured on.  Two families: "decode" (alz_kernels.hip, alz_big.hip and every header they include) and "encode" (alz_encode.hip and its headers; it shares
alz_device.h / alz_internal.h).  sha256 over the files' bytes in name order.  --stamp writes the current hashes into both JSON files (`_kernel_hash`): run it on the
checkout the counters were collected from (tools/profile_r06.sh does), never to silence bench.py's `counters_stale`."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "auroralib", "compression_amd", "csrc")
FAMILIES = {
    "decode": ["alz_kernels.hip", "alz_big.hip", "alz_decode_fast.h", "alz_decode_serial.h", "alz_device.h", "alz_emit_byte.h", "alz_emit_chunk.h", "alz_prs_table.h", "alz_internal.h"],
    "encode": ["alz_encode.hip", "alz_encode_big.h", "alz_encode_seg.h", "alz_encode_seg_seq.h", "alz_device.h", "alz_internal.h"],
}
FILES = ("traffic.json", "insts.json")


def family_files(fam):
    """The family's list, plus any csrc header / kernel file that is in NO list (a new file must not escape the hash): those count for both."""
    listed = set(sum(FAMILIES.values(), []))
    extra = sorted(f for f in os.listdir(CSRC) if (f.endswith(".h") or f.endswith(".hip")) and f not in listed)
    return sorted(FAMILIES[fam]) + extra


def kernel_hash(fam):
    h = hashlib.sha256()
    for f in family_files(fam):
        p = os.path.join(CSRC, f)
        h.update(f.encode() + b"\0")
        if os.path.exists(p):
            with open(p, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def current():
    return {fam: kernel_hash(fam) for fam in FAMILIES}


def recorded(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name))).get("_kernel_hash") or {}
    except Exception:
        return {}


def stale(name, fam):
    """True when profiles/<name> was measured on other kernel sources than the tree's (or carries no hash)."""
    return recorded(name).get(fam) != kernel_hash(fam)


def stamp(families=("decode", "encode")):
    for name in FILES:
        p = os.path.join(ROOT, "profiles", name)
        d = json.load(open(p))
        h = dict(d.get("_kernel_hash") or {})
        for fam in families:
            h[fam] = kernel_hash(fam)
        d["_kernel_hash"] = h
        with open(p, "w") as fh:
            json.dump(d, fh, indent=1)
            fh.write("\n")


if __name__ == "__main__":
    if "--stamp" in sys.argv:
        fams = [a for a in sys.argv[1:] if a in FAMILIES] or list(FAMILIES)
        stamp(fams)
    print(json.dumps({"current": current(), **{n: recorded(n) for n in FILES}}))
