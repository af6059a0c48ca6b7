#!/bin/bash
# Builds libauroralz.so (HIP kernels for gfx950 + C-ABI host code) in-tree.
# hipcc cross-compiles without a GPU.  The HIP runtime is linked by soname (libamdhip64.so.7);
# RUNPATH lists torch's bundled copy first so that a Python host sharing a process with torch
# ends up with ONE runtime (auroralib.compression_amd._lib preloads it), then /opt/rocm/lib.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../../.." && pwd)"
OUT="$HERE/../libauroralz.so"
ROCM="${ROCM_PATH:-/opt/rocm}"
TORCH_LIB="$(python3 - <<'PY' 2>/dev/null || true
import importlib.util, os
s = importlib.util.find_spec("torch")
print(os.path.join(os.path.dirname(s.origin), "lib") if s else "")
PY
)"
FLAGS="-O3 -fPIC --offload-arch=gfx950 -std=c++17 -I$ROOT/include -I$HERE -Wall -Wno-unused-function -Wno-inline-asm"
mkdir -p "$HERE/_obj"
for f in alz_kernels.hip alz_encode.hip alz_big.hip alz_host.cpp alz_container.cpp; do
  [ -f "$HERE/$f" ] || continue
  o="$HERE/_obj/${f%.*}.o"
  if [ ! -f "$o" ] || [ "$HERE/$f" -nt "$o" ] || [ -n "$(find "$HERE" "$ROOT/include" -maxdepth 1 -name '*.h' -newer "$o" 2>/dev/null)" ]; then
    "$ROCM/bin/hipcc" $FLAGS -x hip -c "$HERE/$f" -o "$o" ${ALZ_EXTRA_FLAGS:-}
  fi
  OBJS="${OBJS:-} $o"
done
RP="-Wl,-rpath,$ROCM/lib"
[ -n "$TORCH_LIB" ] && RP="-Wl,-rpath,$TORCH_LIB $RP"
g++ -shared -o "$OUT" $OBJS -L"$ROCM/lib" -lamdhip64 $RP -Wl,--no-undefined -lpthread
echo "built $OUT"
