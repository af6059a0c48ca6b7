cd $GRAFT_REPO_ROOT
for q in 8 0; do echo "== pmc q$q"; bash tools/pmc_encode.sh $q 2>&1 | grep -E "enc_"; done
