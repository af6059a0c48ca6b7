"""Hand-assembled known-answer tests (tests/golden/kat_<format>.json, made by tests/golden/make_kats.py WITHOUT any decoder):
the expected bytes follow from the cited C# token layouts alone, so they pin the oracle's decoders absolutely -- the
round-trip matrix only pins "decoder inverts encoder".  CPU suite: the oracle (both its window models) against the vectors;
-m gpu: the HIP path through the C ABI, both kernel families."""
import base64
import glob
import importlib.util
import json
import os
import zlib

import pytest

import oracle_lib as O
from auroralib.compression_amd import _abi as A

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _cases():
    out = []
    for f in sorted(glob.glob(os.path.join(GOLD, "kat_*.json"))):
        if f.endswith("kat_containers.json"):
            continue
        for c in json.load(open(f))["cases"]:
            out.append(pytest.param(c, id="%s: %s" % (c["format"], c["name"][:60])))
    return out


def _lz(c):
    if "lz" not in c:
        return None
    z = c["lz"]
    lz = A.LzProperties()
    lz.window_bits, lz.length_bits, lz.min_length, lz.windows_start, lz.max_distance = z["window_bits"], z["length_bits"], z["min_length"], z["windows_start"], z["max_distance"]
    return lz


def _expect(c):
    e = zlib.decompress(base64.b64decode(c["expect_zlib_b64"]))
    assert len(e) == c["expect_len"]
    return e


def test_every_north_star_body_has_at_least_three_vectors():
    names = {os.path.basename(f)[4:-5] for f in glob.glob(os.path.join(GOLD, "kat_*.json"))} - {"containers"}
    north = {"lzss", "lz10", "lz11", "yaz0", "yay0", "mio0", "prs_be", "prs_le", "lz4_block", "lzo", "snappy_raw"}
    more = {"clz0", "lz02", "lz40", "lzhudson", "smsr00", "fastlz", "cns", "wflz", "wflz_be", "cnx2", "refpack", "lzshrek", "hig"}   # the other flag-byte formats of the lane-parallel kernel family; FastLZ levels 1 and 2
    assert names == north | more
    for f in glob.glob(os.path.join(GOLD, "kat_*.json")):
        name = os.path.basename(f)[4:-5]
        assert len(json.load(open(f))["cases"]) >= (3 if name in north or name == "containers" else 2), f


def _format_class(c):
    from auroralib.compression_amd import formats as F
    if "cls" in c:
        return getattr(F, c["cls"])
    return {"LZ10": F.LZ10, "LZ11": F.LZ11, "YAZ0": F.Yaz0, "YAY0": F.Yay0, "MIO0": F.MIO0, "LZSS": F.LZSS}[c["container"]]


def _containers():
    return [pytest.param(c, id="%s: %s" % (c["container"], c["name"][:50])) for c in json.load(open(os.path.join(GOLD, "kat_containers.json")))["cases"]]


@pytest.mark.parametrize("c", _containers())
def test_header_layer_of_the_product_reads_hand_built_headers(c):
    """IsMatch / GetDecompressedSize of the C ABI are host code (no GPU needed): checked against headers the library never
    wrote -- product and oracle share the mould of their header layers, these bytes come from neither."""
    f = _format_class(c)()
    blob = bytes.fromhex(c["file"])
    if "zero u24" not in c["name"] and not c.get("skip_is_match"):   # (LZ10.Validate wants a plausible u24 size)
        assert f.IsMatch(blob) == c.get("is_match", True)
    if "little-endian size" not in c["name"] and c.get("provides_size", True):   # (GetDecompressedSize does not retry; Decompress does)
        assert f.GetDecompressedSize(blob) == c["expect_len"]


@pytest.mark.parametrize("c", _containers())
def test_oracle_container_layer_reads_hand_built_files(c):
    blob = bytes.fromhex(c["file"])
    out, st = O.container_decompress(getattr(A, "C_" + c["container"]), blob, big_endian=bool(c["big_endian"]))
    assert st == 0 and out == _expect(c)


@pytest.mark.gpu
@pytest.mark.parametrize("c", _containers())
def test_gpu_container_decompress_of_hand_built_files(c):
    assert _format_class(c)().Decompress(bytes.fromhex(c["file"])) == _expect(c)


def test_committed_vectors_are_what_the_generator_writes(tmp_path):
    """The generator is the derivation: it must reproduce the committed files byte for byte (and it imports nothing of ours)."""
    src = open(os.path.join(GOLD, "make_kats.py")).read()
    assert "import oracle" not in src and "oracle_lib" not in src and "auroralib" not in src and "ctypes" not in src and "subprocess" not in src
    spec = importlib.util.spec_from_file_location("make_kats", os.path.join(GOLD, "make_kats.py"))
    mk = importlib.util.module_from_spec(spec); spec.loader.exec_module(mk)
    for fmt, cases in mk.build().items():
        assert json.load(open(os.path.join(GOLD, "kat_%s.json" % fmt)))["cases"] == json.loads(json.dumps(cases)), fmt
    assert json.load(open(os.path.join(GOLD, "kat_containers.json")))["cases"] == json.loads(json.dumps(mk.build_containers()))


@pytest.mark.parametrize("c", _cases())
@pytest.mark.parametrize("flat", [False, True], ids=["ring+flush", "flat"])
def test_oracle_decodes_the_vector(c, flat):
    exp = _expect(c)
    fmt = A.FORMAT_NAMES.index(c["format"])
    out, r = O.decode_stream(fmt, bytes.fromhex(c["src"]), decom_len=c["decom_len"], cap=len(exp), aux0=c["aux0"], aux1=c["aux1"], lz=_lz(c), flat=flat)
    assert r.status == A.ST_OK and r.dst_len == len(exp), (r.status, r.dst_len, len(exp))
    assert out == exp
    if "src_used" in c:
        assert r.src_used == c["src_used"]


@pytest.mark.gpu
@pytest.mark.parametrize("c", _cases())
def test_gpu_decodes_the_vector(c):
    from gpu_common import ctx
    exp = _expect(c)
    fmt = A.FORMAT_NAMES.index(c["format"])
    for exact in (1, 0):
        ctx().set_exact_kernels(exact)
        try:
            out, r = ctx().decode(fmt, bytes.fromhex(c["src"]), decom_len=c["decom_len"], cap=len(exp), aux0=c["aux0"], aux1=c["aux1"], lz=_lz(c))
        finally:
            ctx().set_exact_kernels(0)
        assert r.status == A.ST_OK and r.dst_len == len(exp), (exact, r.status, r.dst_len, len(exp))
        assert out == exp, "kernel family %d" % exact
        if "src_used" in c:
            assert r.src_used == c["src_used"]
