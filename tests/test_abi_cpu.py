"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol declared in
include/auroralz.h, struct layouts match, and the product path refuses to run without a GPU."""
import ctypes as C
import os
import re

import pytest

from auroralib.compression_amd import _abi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "auroralz.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(alz_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from auroralib.compression_amd import _lib
    lib = _lib.load()
    syms = _declared_symbols()
    assert len(syms) >= 20
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.alz_abi_version() == A.ABI_VERSION


def test_struct_layouts():
    assert C.sizeof(A.Stream) == 40 and A.Stream.format.offset == 36
    assert C.sizeof(A.Result) == 16 and C.sizeof(A.LzProperties) == 16 and C.sizeof(A.Settings) == 16
    lz = A.LzProperties.from_bits(12, 4, 2)
    assert (lz.window_bits, lz.length_bits, lz.min_length, lz.windows_start, lz.max_distance) == (12, 4, 3, 0xFEE, 4096)
    lz = A.LzProperties.from_bits(10, 6, 2)
    assert (lz.windows_start, lz.max_distance) == (958, 1024)


def test_single_hip_runtime_in_process():
    from auroralib.compression_amd import _lib
    _lib.load()
    maps = {line.split()[-1] for line in open("/proc/self/maps") if "libamdhip64" in line}
    assert len(maps) == 1, maps


def test_no_cpu_fallback_without_device():
    """Without a GPU the product path must fail loudly, never route through a CPU decoder."""
    from auroralib.compression_amd import _lib
    lib = _lib.load()
    if lib.alz_device_count() > 0:
        pytest.skip("a GPU is present")
    from auroralib.compression_amd.batch import Context
    with pytest.raises(_lib.AlzError) as ei:
        Context(0)
    assert ei.value.code == A.E_NO_DEVICE


def test_product_does_not_reference_oracle():
    """The oracle is test infrastructure: nothing under auroralib/ may import, link or dlopen it."""
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "auroralib")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".c", ".sh")):
                t = open(os.path.join(dp, f), errors="replace").read()
                if "liboracle" in t or "alz_oracle" in t or "oracle_lib" in t:
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
    import subprocess
    so = os.path.join(ROOT, "auroralib", "compression_amd", "libauroralz.so")
    needed = subprocess.run(["objdump", "-p", so], capture_output=True, text=True).stdout
    assert "oracle" not in needed


def test_header_is_plain_c_and_links(tmp_path):
    """include/auroralz.h is the contract a C / C# / Go host binds: it must compile as strict C99 and the symbols must
    link from plain C (no C++ name mangling, no torch or HIP types in the signatures)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "host.c"
    src.write_text(r"""
#include <stdio.h>
#include "auroralz.h"
int main(void) {
    alz_stream s; alz_result r; alz_container_options o; alz_scan_hit h;
    (void)s; (void)r; (void)o; (void)h;
    if (sizeof(alz_stream) != 40 || sizeof(alz_result) != 16 || sizeof(alz_lz_properties) != 16) return 2;
    printf("%d %d %s\n", alz_abi_version(), (int)ALZ_C_COUNT, alz_brute_decoder_name(4));
    return alz_container_is_match(ALZ_C_YAZ0, (const uint8_t*)"Yaz0\0\0\1\0\0\0\0\0\0\0\0\0xxxxxxxx", 24) == 1 ? 0 : 3;
}
""")
    exe = tmp_path / "host"
    libdir = os.path.join(root, "auroralib", "compression_amd")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-lauroralz", "-Wl,-rpath," + libdir])
    out = subprocess.check_output([str(exe)]).decode().split()
    assert out[0] == str(A.ABI_VERSION) and out[1] == str(A.C_COUNT) and " ".join(out[2:]) == "LZSS (10, 6, 2)"


def test_decode_kernels_use_no_scratch(tmp_path):
    """Every decode kernel keeps its state in registers and LDS: a non-zero private segment means a local ended up in
    scratch memory (it did once: store sinking through a pointer phi put the Yay0 / MIO0 cursors there, -25 %)."""
    import re
    import shutil
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    if not (os.path.exists(llvm + "/llvm-objdump") and os.path.exists(llvm + "/llvm-readelf")):
        pytest.skip("no llvm binutils")
    so = shutil.copy(os.path.join(ROOT, "auroralib", "compression_amd", "libauroralz.so"), tmp_path / "lib.so")
    subprocess.run([llvm + "/llvm-objdump", "--offloading", str(so)], check=True, capture_output=True, cwd=tmp_path)
    seen = {}
    for co in tmp_path.glob("lib.so.*gfx950"):
        notes = subprocess.run([llvm + "/llvm-readelf", "--notes", str(co)], check=True, capture_output=True, text=True).stdout
        for name, priv in re.findall(r"\.name:\s+(\S+)\n\s+\.private_segment_fixed_size:\s+(\d+)", notes):
            seen[name] = int(priv)
        for name, spills in re.findall(r"\.name:\s+(\S+)\n(?:\s+\.\w+:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", notes):
            assert int(spills) == 0 or "alz_decode_" not in name, (name, spills)
    dec = {k: v for k, v in seen.items() if "alz_decode_" in k}
    assert len(dec) >= 25, sorted(seen)
    # The queue kernels of the 64 KiB formats call the chunked byte phase out of line from the exact parsers' token sites
    # (queue_emit_call, one copy per kernel: inlining it at every site made 100 KB kernels); that callee saves ONE
    # callee-saved VGPR of the calling convention on its frame -- 8 bytes, outside every loop.  Anything beyond that is a local in
    # scratch memory.
    assert all(v == 0 or (v <= 8 and ("alz_decode_queue_kernel" in k or "alz_decode_queue2_kernel" in k)) for k, v in dec.items()), {k: v for k, v in dec.items() if v}


def test_lds_table_kernel_fits_one_cu(tmp_path):
    """Kernel A of the encoder (enc_prev_cu_kernel: head table, queues and staging rows in LDS, one workgroup of 1 024 threads per CU)
    must fit the 160 KB of LDS and the 128 registers a wavefront of such a workgroup gets, without spilling."""
    import re
    import shutil
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    if not (os.path.exists(llvm + "/llvm-objdump") and os.path.exists(llvm + "/llvm-readelf")):
        pytest.skip("no llvm binutils")
    so = shutil.copy(os.path.join(ROOT, "auroralib", "compression_amd", "libauroralz.so"), tmp_path / "lib.so")
    subprocess.run([llvm + "/llvm-objdump", "--offloading", str(so)], check=True, capture_output=True, cwd=tmp_path)
    found = 0
    for co in tmp_path.glob("lib.so.*gfx950"):
        notes = subprocess.run([llvm + "/llvm-readelf", "--notes", str(co)], check=True, capture_output=True, text=True).stdout
        for block in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", block)
            if not name or "enc_prev_cu_kernel" not in name.group(1):
                continue
            found += 1
            lds = int(re.search(r"\.group_segment_fixed_size:\s+(\d+)", block).group(1))
            priv = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", block).group(1))
            vgpr = int(re.search(r"\.vgpr_count:\s+(\d+)", block).group(1))
            assert 128 * 1024 < lds <= 160 * 1024 and priv == 0 and vgpr <= 128, (name.group(1), lds, priv, vgpr)
    assert found == 3            # one pass (2 groups of 64 entries per wavefront and chunk), several passes (3), one pass with the tag / link rings (windows up to 8 KiB)


def test_library_reads_no_environment_switch_but_the_two_documented():
    """Kernel selection is an API (alz_ctx_set_exact_kernels / alz_ctx_set_kernel_variant), never the caller's environment: the only
    variables the library looks at are ALZ_COPY_THREADS (host copy threads of the staging path) and ALZ_TIMING (phase times on
    stderr).  Experiment code is compiled in only with -DALZ_EXPERIMENTS, which build.sh never sets."""
    import re
    blob = open(os.path.join(ROOT, "auroralib", "compression_amd", "libauroralz.so"), "rb").read()
    names = set(m.group(0).decode() for m in re.finditer(rb"ALZ_[A-Z0-9_]{3,}", blob))
    env_like = {n for n in names if not n.startswith(("ALZ_E_", "ALZ_ST_", "ALZ_FMT_", "ALZ_C_"))}
    assert env_like <= {"ALZ_COPY_THREADS", "ALZ_TIMING"}, env_like
    for f in ("alz_kernels.hip", "alz_encode.hip", "alz_host.cpp", "alz_container.cpp", "alz_decode_fast.h"):
        text = open(os.path.join(ROOT, "auroralib", "compression_amd", "csrc", f)).read()
        for m in re.finditer(r'getenv\("(\w+)"\)', text):
            assert m.group(1) in ("ALZ_COPY_THREADS", "ALZ_TIMING"), (f, m.group(1))
    assert "ALZ_EXPERIMENTS" not in open(os.path.join(ROOT, "auroralib", "compression_amd", "csrc", "build.sh")).read()
