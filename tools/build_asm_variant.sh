#!/bin/bash
# usage: bash tools/build_asm_variant.sh NAME FILTER.py [source, default alz_kernels.hip]
#   -- an experiment build at the ISA level: the device assembly of one source file is passed through `python3 FILTER.py < in.s > out.s`, assembled,
#   linked and bundled the way hipcc does it, and the library linked with it -> build/variants/NAME.so.  The product tree is not touched.
set -e
cd "$(dirname "$0")/.."
name="$1"; filt="$2"; file="${3:-alz_kernels.hip}"
C=auroralib/compression_amd/csrc; L=/opt/rocm/lib/llvm/bin; T=/tmp/asmvar_$name; mkdir -p $T build/variants
FLAGS="-O3 -fPIC --offload-arch=gfx950 -std=c++17 -Iinclude -I$C -Wno-unused-function -Wno-inline-asm"
/opt/rocm/bin/hipcc $FLAGS -x hip --cuda-device-only -S $C/$file -o $T/dev.s 2>/dev/null
python3 "$filt" < $T/dev.s > $T/dev2.s
$L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $T/dev2.s -o $T/dev.o
$L/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $T/dev.out $T/dev.o
$L/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$T/dev.out -output=$T/dev.hipfb
/opt/rocm/bin/hipcc $FLAGS -x hip --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $T/dev.hipfb -c $C/$file -o $T/host.o 2>/dev/null
OBJS=""
for f in alz_kernels alz_encode alz_big alz_host alz_container; do
  if [ "$f.hip" = "$file" ] || [ "$f.cpp" = "$file" ]; then OBJS="$OBJS $T/host.o"; else OBJS="$OBJS $C/_obj/$f.o"; fi
done
TORCH_LIB="$(python3 -c 'import importlib.util,os; s=importlib.util.find_spec("torch"); print(os.path.join(os.path.dirname(s.origin),"lib"))' 2>/dev/null || true)"
g++ -shared -o build/variants/$name.so $OBJS -L/opt/rocm/lib -lamdhip64 ${TORCH_LIB:+-Wl,-rpath,$TORCH_LIB} -Wl,-rpath,/opt/rocm/lib -Wl,--no-undefined -lpthread
echo "build/variants/$name.so"
