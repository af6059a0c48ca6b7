cd $GRAFT_REPO_ROOT
echo "== gpu suite"; timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -4
echo "== enc profile"; bash tools/enc_profile.sh > /dev/null 2>&1; grep -E "quality|enc_|workload" gpurun_out/r03_encode.md | cut -c1-200
echo "== traffic"; bash tools/traffic_encode.sh 2>&1 | grep -v "^W\|amdgpu.ids"
echo "== bench encode"; timeout 900 python bench.py --mode encode > gpurun_out/bench_encode.json 2> gpurun_out/bench_encode.err; tail -c 2500 gpurun_out/bench_encode.json; tail -3 gpurun_out/bench_encode.err
echo "== formats q8"; bash tools/enc_formats.sh 2>&1 | tail -14
