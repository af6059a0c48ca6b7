"""The managed half of the boundary (shim/AuroraLib.Compression.Amd/*.cs) cannot be compiled here (no .NET SDK), so its
binding surface is checked textually against include/auroralz.h: every [DllImport] must name a function the header declares
with the same number of parameters, and every [StructLayout(Sequential)] struct must have the byte size of its C twin
(sizeof from a C program compiled against the header).  Also: every format class implements the ICompressionAlgorithm surface."""
import glob
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "shim", "AuroraLib.Compression.Amd")
HDR = os.path.join(ROOT, "include", "auroralz.h")

CS_SIZES = {"byte": 1, "sbyte": 1, "short": 2, "ushort": 2, "int": 4, "uint": 4, "long": 8, "ulong": 8, "float": 4, "double": 8}


def _header_prototypes():
    text = re.sub(r"/\*.*?\*/", " ", open(HDR).read(), flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int|void|size_t|const\s+char\s*\*)\s+(alz_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        params = m.group(2).strip()
        n = 0 if params in ("", "void") else len([p for p in params.split(",") if p.strip()])
        protos[m.group(1)] = n
    return protos


def _dllimports():
    out = {}
    src = open(os.path.join(SHIM, "Native.cs")).read()
    for m in re.finditer(r"\[DllImport\(Lib\)\]\s*internal\s+static\s+extern\s+[\w\.]+\*?\s+(\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        params = m.group(2).strip()
        out[m.group(1)] = 0 if not params else len([p for p in params.split(",") if p.strip()])
    return out


def test_every_dllimport_matches_a_header_prototype():
    protos, imports = _header_prototypes(), _dllimports()
    assert len(imports) >= 12, imports
    for name, arity in imports.items():
        assert name in protos, "%s is not declared in include/auroralz.h" % name
        assert protos[name] == arity, "%s: header has %d parameters, Native.cs %d" % (name, protos[name], arity)
    # the entry points the shim is built on are all bound
    for need in ("alz_create", "alz_destroy", "alz_decode", "alz_decode_batch", "alz_decode_batch_multi", "alz_encode_batch", "alz_last_error"):
        assert need in imports


def test_struct_sizes_match_the_c_abi(tmp_path):
    src = open(os.path.join(SHIM, "Native.cs")).read()
    cs = {}
    for m in re.finditer(r"\[StructLayout\(LayoutKind\.Sequential\)\]\s*public\s+struct\s+(\w+)\s*\{(.*?)\n    \}", src, flags=re.S):
        size = 0
        for f in re.finditer(r"public\s+(\w+)\s+([\w,\s]+);", m.group(2)):
            size += CS_SIZES[f.group(1)] * len([x for x in f.group(2).split(",") if x.strip()])
        cs[m.group(1)] = size
    twins = {"AlzLzProperties": "alz_lz_properties", "AlzStream": "alz_stream", "AlzResult": "alz_result", "AlzSettings": "alz_settings", "AlzEncodeAux": "alz_encode_aux"}
    assert set(cs) == set(twins), cs
    prog = tmp_path / "sz.c"
    prog.write_text('#include <stdio.h>\n#include "auroralz.h"\nint main(void){' + "".join('printf("%s %%zu\\n", sizeof(%s));' % (k, v) for k, v in twins.items()) + "return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(prog)])
    c = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
    for k in twins:
        assert int(c[k]) == cs[k], "%s: C %s bytes, C# fields sum to %d" % (k, c[k], cs[k])   # (fields are ordered so that no padding is needed)


def test_format_and_status_enums_match_the_header():
    hdr = open(HDR).read()
    src = open(os.path.join(SHIM, "Native.cs")).read()
    fm = dict((m.group(1), int(m.group(2))) for m in re.finditer(r"ALZ_FMT_(\w+)\s*=\s*(\d+)", hdr))
    body = re.search(r"public enum AlzFormat : uint\s*\{(.*?)\}", src, flags=re.S).group(1)
    cs = dict((m.group(1).upper().replace("_", ""), int(m.group(2))) for m in re.finditer(r"(\w+)\s*=\s*(\d+)", body))
    want = {k.replace("_", ""): v for k, v in fm.items() if k != "COUNT"}
    want = {k.replace("PRSBE", "PRSBE").replace("LZ4BLOCK", "LZ4BLOCK").replace("SNAPPYRAW", "SNAPPYRAW"): v for k, v in want.items()}
    assert cs == want
    assert re.search(r"Snappy = 9, LZ4Frame = 22", src) and "ALZ_C_SNAPPY = 9" in hdr.replace("  ", " ") and re.search(r"ALZ_C_LZ4_FRAME = 22", hdr)
    assert "_available = Native.alz_abi_version() == %s" % re.search(r"#define ALZ_ABI_VERSION (\d+)", hdr).group(1) in open(os.path.join(SHIM, "AmdContext.cs")).read()


def test_every_format_class_implements_the_reference_surface():
    """ICompressionAlgorithm = ICompressionDecoder (Info, IsMatch, Decompress) + ICompressionEncoder (Compress)
    (src/AuroraLib.Compression/Interfaces/ICompressionAlgorithm.cs:6); size / byte-order interfaces as the mirrored class has them."""
    want = {"LZ10": True, "LZ11": True, "Yaz0": True, "Yay0": True, "MIO0": True, "LZSS": True, "PRS": False, "LZO": False, "LZ4": False, "Snappy": False}
    text = "\n".join(open(f).read() for f in glob.glob(os.path.join(SHIM, "*.cs")))
    for cls, sized in want.items():
        m = re.search(r"public (?:sealed )?class %s : ([^\n{]+)\s*\{(.*?)\n    \}" % cls, text, flags=re.S)
        assert m, cls
        ifaces, body = m.group(1), m.group(2)
        assert "ICompressionAlgorithm" in ifaces
        for member in (r"public (?:virtual )?IFormatInfo Info", r"public (?:virtual )?bool IsMatch\(Stream stream, ReadOnlySpan<char> fileNameAndExtension = default\)",
                       r"public (?:virtual )?void Decompress\(Stream source, Stream destination\)",
                       r"public (?:virtual )?void Compress\(ReadOnlySpan<byte> source, Stream destination, CompressionSettings settings = default\)"):
            assert re.search(member, body), (cls, member)
        assert ("IProvidesDecompressedSize" in ifaces) == sized and (re.search(r"public uint GetDecompressedSize\(Stream source\)", body) is not None) == sized, cls
        assert re.search(r"new FormatInfo<%s>\(" % cls, body), cls        # discovered by reflection: own FormatInfo, parameterless ctor
    for cls in ("Yaz0", "Yay0", "MIO0", "PRS"):
        assert re.search(r"class %s : [^\n{]*IEndianDependentFormat" % cls, text) and re.search(r"public Endian FormatByteOrder \{ get; set; \}", text)
    code = re.sub(r"//[^\n]*", "", text)
    assert ".ReadExactly(" not in code and not re.search(r"\bnuint\b", code)     # netstandard2.0 / net472 targets of the reference


# ------------------------------------------------------------------------------------------------------------------
# Behaviour the compiler would not catch either: property defaults, static signatures with their default arguments and the
# interface lists of every shim class against the reference class it names (needs /root/reference: skipped on the GPU box),
# and the parameter TYPES of every [DllImport] against the header.
import pytest

REF = "/root/reference/src"
REF_CLASSES = {   # shim file -> (shim class, reference files that hold the class)
    "LZ10.cs": ("LZ10", ["AuroraLib.Compression.Nintendo/Nintendo/LZ10.cs"]),
    "LZ11.cs": ("LZ11", ["AuroraLib.Compression.Nintendo/Nintendo/LZ11.cs"]),
    "Yaz0.cs": ("Yaz0", ["AuroraLib.Compression.Nintendo/Nintendo/Yaz0.cs"]),
    "Yay0.cs": ("Yay0", ["AuroraLib.Compression.Nintendo/Nintendo/Yay0.cs"]),
    "MIO0.cs": ("MIO0", ["AuroraLib.Compression.Nintendo/Nintendo/MIO0.cs"]),
    "LZSS.cs": ("LZSS", ["AuroraLib.Compression/Formats/Common/LZSS.cs"]),
    "LZO.cs": ("LZO", ["AuroraLib.Compression/Formats/Common/LZO.cs"]),
    "PRS.cs": ("PRS", ["AuroraLib.Compression.Sega/Sega/PRS.cs"]),
    "Framed.cs:LZ4": ("LZ4", ["AuroraLib.Compression/Formats/Common/LZ4.cs", "AuroraLib.Compression/Formats/Common/LZ4.Frame.cs"]),
    "Framed.cs:Snappy": ("Snappy", ["AuroraLib.Compression/Formats/Common/Snappy.cs"]),
}


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def _class_body(text, cls):
    m = re.search(r"public (?:sealed |partial |static )*class %s\b\s*(?::\s*([^\n{]+))?\s*\{" % cls, text)
    assert m, cls
    depth, i = 1, m.end()
    while depth and i < len(text):
        depth += {"{": 1, "}": -1}.get(text[i], 0)
        i += 1
    return (m.group(1) or ""), text[m.end():i - 1]


def _norm_params(params):
    out = []
    for p in re.split(r",(?![^<]*>)", params):
        p = " ".join(p.split())
        if p:
            out.append(p)
    return tuple(out)


def _surface(text, cls):
    ifaces, body = _class_body(_strip_comments(text), cls)
    ifaces = set(x.strip().split(".")[-1] for x in ifaces.split(",") if x.strip())
    props = {}
    for m in re.finditer(r"public (?:static |virtual |override )*([\w<>\.\?\[\]]+) (\w+)\s*\{\s*get;\s*set;\s*\}\s*(?:=\s*([^;]+);)?", body):
        props[m.group(2)] = (m.group(1), (m.group(3) or "").strip())
    statics = {}
    for m in re.finditer(r"public static (?:unsafe )?([\w<>\.\?\[\]]+) (\w*(?:DecompressHeaderless|CompressHeaderless|CompressBlockHeaderless|DecompressBlockHeaderless))\s*\(([^)]*)\)", body):
        statics.setdefault(m.group(2), set()).add((m.group(1), _norm_params(m.group(3))))
    return ifaces, props, statics


ZERO_DEFAULTS = {"", "0", "default", "false", "null"}


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree only exists in the build container")
@pytest.mark.parametrize("key", sorted(REF_CLASSES))
def test_shim_class_mirrors_the_reference_class(key):
    cls, ref_files = REF_CLASSES[key]
    shim_text = open(os.path.join(SHIM, key.split(":")[0])).read()
    ref_text = "\n".join(open(os.path.join(REF, f), encoding="utf-8-sig").read() for f in ref_files)
    s_if, s_props, s_stat = _surface(shim_text, cls)
    r_if, r_props, r_stat = set(), {}, {}
    for f in ref_files:
        i, p, st = _surface(open(os.path.join(REF, f), encoding="utf-8-sig").read(), cls)
        r_if |= i; r_props.update(p)
        for k, v in st.items():
            r_stat.setdefault(k, set()).update(v)
    assert ref_text
    # interfaces: the shim class implements exactly what the class it replaces implements
    assert s_if == r_if, (cls, s_if, r_if)
    # every public settable property of the reference class exists with the same type and the same default
    for name, (typ, dflt) in r_props.items():
        assert name in s_props, "%s.%s is missing in the shim" % (cls, name)
        st, sd = s_props[name]
        assert st == typ, (cls, name, st, typ)
        assert sd == dflt or (sd in ZERO_DEFAULTS and dflt in ZERO_DEFAULTS), "%s.%s defaults to %r, the reference to %r" % (cls, name, sd, dflt)
    assert set(s_props) <= set(r_props), (cls, set(s_props) - set(r_props))
    # every static body entry point the shim offers has a twin in the reference: same name, return type, parameter types, names
    # AND default arguments
    for name, overloads in s_stat.items():
        assert name in r_stat, "%s.%s does not exist in the reference" % (cls, name)
        for ov in overloads:
            assert ov in r_stat[name], "%s.%s%r: the reference has %r" % (cls, name, ov, sorted(r_stat[name]))
    if cls not in ("LZ4", "Snappy", "Yay0", "MIO0"):       # (framed files and the three-section formats go through Compress(); their statics take separate streams)
        assert "CompressHeaderless" in s_stat and "DecompressHeaderless" in s_stat, cls
        for name in ("CompressHeaderless", "DecompressHeaderless"):
            assert len(s_stat[name]) == len(r_stat[name]), "%s.%s: %d overloads, the reference has %d" % (cls, name, len(s_stat[name]), len(r_stat[name]))


C_TO_CS = [   # (regex over a normalised C parameter type, C# parameter type)
    (r"alz_ctx\*\*", "out IntPtr"), (r"alz_ctx\* ?const\*", "IntPtr*"), (r"alz_ctx\*", "IntPtr"),
    (r"const alz_lz_properties\*", "AlzLzProperties*"), (r"const alz_settings\*", "AlzSettings*"), (r"const alz_stream\*", "AlzStream*"),
    (r"alz_result\*", "AlzResult*"), (r"alz_encode_aux\*", "AlzEncodeAux*"), (r"const alz_container_options\*", "void*"),
    (r"(?:const )?uint8_t\*", "byte*"), (r"uint32_t\*", "uint*"), (r"uint64_t\*", "ulong*"), (r"int32_t\*|int\*", "int*"), (r"size_t\*", "UIntPtr*"),
    (r"uint32_t", "uint"), (r"uint64_t", "ulong"), (r"size_t", "UIntPtr"), (r"int32_t|int", "int"),
]


def _c_param_types(params):
    out = []
    for p in params.split(","):
        p = " ".join(re.sub(r"/\*.*?\*/", " ", p).split())
        if not p or p == "void":
            continue
        t = re.sub(r"\s*\b\w+$", "", p) if not p.endswith("*") else p      # drop the parameter name
        t = t.replace(" *", "*").replace("* ", "*").strip()
        out.append(t)
    return out


def test_dllimport_parameter_types_match_the_header():
    text = re.sub(r"/\*.*?\*/", " ", open(HDR).read(), flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|void|size_t|const\s+char\s*\*)\s+(alz_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        protos[m.group(2)] = (" ".join(m.group(1).split()), _c_param_types(m.group(3)))
    src = open(os.path.join(SHIM, "Native.cs")).read()
    n = 0
    for m in re.finditer(r"\[DllImport\(Lib\)\]\s*internal\s+static\s+extern\s+([\w\.]+\*?)\s+(\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        ret, name, params = m.group(1), m.group(2), m.group(3)
        c_ret, c_types = protos[name]
        want_ret = {"int": "int", "void": "void", "size_t": "UIntPtr", "const char*": "IntPtr", "const char *": "IntPtr"}[c_ret]
        assert ret == want_ret, (name, ret, want_ret)
        cs_types = [" ".join(p.split()).rsplit(" ", 1)[0] for p in params.split(",") if p.strip()]
        assert len(cs_types) == len(c_types), name
        for ct, cst in zip(c_types, cs_types):
            for rx, want in C_TO_CS:
                if re.fullmatch(rx, ct):
                    assert cst == want, "%s: C parameter %r is bound as %r, expected %r" % (name, ct, cst, want)
                    break
            else:
                raise AssertionError("%s: no C# mapping for C type %r" % (name, ct))
            n += 1
    assert n > 60


def test_single_stream_compress_stays_managed_by_default():
    """One buffer is a serial job for one wavefront: Compress / CompressHeaderless reach the native encoder only above
    AmdContext.SingleStreamCompressThreshold (default: never) -- BatchEncoder.CompressMany is the GPU entry point -- and a
    caller's MaxWindowBits beyond the format's own window (where the managed finder returns distances the format cannot store, and
    alz_encode_batch answers ALZ_E_UNSUPPORTED) stays with the managed encoder; the shim's rule mirrors the library's per format."""
    ctx = open(os.path.join(SHIM, "AmdContext.cs")).read()
    assert re.search(r"public static uint SingleStreamCompressThreshold \{ get; set; \} = uint\.MaxValue;", ctx)
    body = open(os.path.join(SHIM, "AmdBody.cs")).read()
    assert "MaxWindowBitsOnGpu(format, settings.MaxWindowBits, lz)" in body and "SingleStreamCompressThreshold" in body
    assert "if (format == AlzFormat.FastLZ) return bits <= 20;" in body and "return bits <= wb && (1 << bits) <= maxDistance;" in body
    # the window table of the shim against the library's own (alz_encode.hip: alz_encode_geometry)
    hip = open(os.path.join(ROOT, "auroralib", "compression_amd", "csrc", "alz_encode.hip")).read()
    for name, wb, md in (("PRS_BE", 13, "0x1FFF"), ("LZ4_BLOCK", 16, "0xFFFF"), ("LZO", 16, "0xBFFF"), ("SNAPPY_RAW", 15, "0x8000")):
        assert re.search(r"case ALZ_FMT_%s:[^\n]*wb = %d;[^\n]*g\.max_dist = %s;" % (name, wb, md), hip), name
        assert "(%d, %s)" % (wb, md) in body, name
    for f in ("LZ10.cs", "LZ11.cs", "Yaz0.cs", "Yay0.cs", "MIO0.cs", "LZSS.cs", "LZO.cs", "PRS.cs"):
        t = _strip_comments(open(os.path.join(SHIM, f)).read())
        assert "AmdBody.UseGpuForCompress(" in t, f
        assert not re.search(r"if \(!AmdContext\.Available\)", t), f     # no unconditional "a device exists -> GPU"
    enc = open(os.path.join(SHIM, "BatchEncoder.cs")).read()
    assert "public static EncodedBody[] CompressMany(" in enc and "Native.alz_encode_batch(" in enc
