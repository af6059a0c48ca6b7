cd "${GRAFT_REPO_ROOT:?}" || exit 1
patch -p1 -N -s < tools/variants/r04_encode_switches.patch || true   # (the compile-time switches this script turns live in a patch, not in the product sources; the GPU box works on a scratch copy)
export ALZ_SINGLE_MODES=big ALZ_SINGLE_Q=4,8,12
for fl in "" "-DALZ_BENC_DENSE=1"; do
  touch auroralib/compression_amd/csrc/alz_encode.hip
  ALZ_EXTRA_FLAGS="$fl" bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
  echo "== flags '$fl' text"
  ALZ_SINGLE_DATA=text timeout 600 python tools/single_encode.py lzss yaz0 lz4_block 2>&1 | grep -v amdgpu | cut -c1-100
done
touch auroralib/compression_amd/csrc/alz_encode.hip; bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
