# usage (GPU box): bash tools/occ_pad.sh "<pad bytes...>" [formats...] -- one batch in flight against extra dynamic LDS per wave (resident waves per CU)
cd $GRAFT_REPO_ROOT
pads="$1"; shift
for p in $pads; do
  echo "pad $p:"
  ALZ_OCC_PAD=$p bash tools/ab.sh "$@"
done
