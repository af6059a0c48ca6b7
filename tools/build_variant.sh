#!/bin/bash
# usage: bash tools/build_variant.sh name "-DFLAG=1 ..." [source file to rebuild, default alz_encode.hip]  -- builds the library with extra compiler flags into build/variants/name.so
# (experiments: on the GPU box copy a variant over auroralib/compression_amd/libauroralz.so before running a tool)
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants
cp auroralib/compression_amd/libauroralz.so /tmp/libauroralz.keep 2>/dev/null || true
touch auroralib/compression_amd/csrc/${3:-alz_encode.hip}
ALZ_EXTRA_FLAGS="$2" bash auroralib/compression_amd/csrc/build.sh > /dev/null
cp auroralib/compression_amd/libauroralz.so build/variants/$1.so
touch auroralib/compression_amd/csrc/${3:-alz_encode.hip}
echo "build/variants/$1.so"
