# usage (GPU box): bash tools/r3_prs.sh  -- PRS parity tests, then kernel ms of PRS (product build and variants under build/variants/prs_*.so)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "prs or PRS or mixed or kat" 2>&1 | tail -5
bash tools/ab.sh prs_be prs_le yaz0
cp auroralib/compression_amd/libauroralz.so /tmp/keep.so
for v in build/variants/prs_*.so; do
  [ -f "$v" ] || continue
  cp $v auroralib/compression_amd/libauroralz.so
  echo "== $v"; bash tools/ab.sh prs_be
done
cp /tmp/keep.so auroralib/compression_amd/libauroralz.so
