#!/bin/bash
# usage: bash tools/build_variant.sh name "-DFLAG=1 ..." [source file to rebuild, default alz_encode.hip] [--patch tools/variants/x.patch]
#   -- builds the library with extra compiler flags (and, optionally, an experiment patch applied for the duration of the build) into
#   build/variants/name.so; the product tree and libauroralz.so are put back afterwards
# (experiments: on the GPU box copy a variant over auroralib/compression_amd/libauroralz.so before running a tool -- tools/variants.sh)
set -e
cd "$(dirname "$0")/.."
name="$1"; flags="$2"; file="${3:-alz_encode.hip}"; patch=""
if [ "$4" = "--patch" ]; then patch="$5"; fi
mkdir -p build/variants
cp auroralib/compression_amd/libauroralz.so /tmp/libauroralz.keep 2>/dev/null || true
if [ -n "$patch" ]; then git apply "$patch"; fi
restore() { if [ -n "$patch" ]; then git apply -R "$patch"; fi; touch auroralib/compression_amd/csrc/*.hip; cp /tmp/libauroralz.keep auroralib/compression_amd/libauroralz.so 2>/dev/null || true; }
trap restore EXIT
touch auroralib/compression_amd/csrc/*.hip
ALZ_EXTRA_FLAGS="$flags" bash auroralib/compression_amd/csrc/build.sh > /dev/null
cp auroralib/compression_amd/libauroralz.so build/variants/$name.so
echo "build/variants/$name.so"
