cd $GRAFT_REPO_ROOT
echo "== fuzz soak"; bash tools/soak.sh 9100 12 2>&1 | tail -14
echo "== synth soak"; timeout 1200 python tools/soak_synth.py 4100 6 2>&1 | tail -6
echo "== encode soak"; timeout 1500 python tools/soak_encode.py 8100 8 2>&1 | tail -8
