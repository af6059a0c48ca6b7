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
