#!/bin/bash
# usage (GPU box): bash tools/variants.sh "tool command" name1 name2 ...  -- runs the command once per library variant of build/variants
cmd="$1"; shift
cp auroralib/compression_amd/libauroralz.so /tmp/lib.keep
for v in "$@"; do
  echo "== variant $v"
  cp build/variants/$v.so auroralib/compression_amd/libauroralz.so
  eval "$cmd"
done
cp /tmp/lib.keep auroralib/compression_amd/libauroralz.so
