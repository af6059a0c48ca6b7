# usage (GPU box): bash tools/exp_flags.sh "<extra hipcc flags>" <bench args...>  -- rebuild with the flags, run bench.py, print the
# headline + configs values, rebuild the product library
cd $GRAFT_REPO_ROOT
FLAGS=$1; shift
rm -rf auroralib/compression_amd/csrc/_obj
ALZ_EXTRA_FLAGS="$FLAGS" bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('flags [$FLAGS]:', d['config']['format'], d['value'], 'kernel_ms', d['roofline']['kernel_ms'], d['config']['parity_ok'])
for c in d.get('configs') or []: print('   ', c.get('name'), c.get('value'), c.get('roofline',{}).get('kernel_ms'), c.get('parity_ok'), c.get('error',''))"
rm -rf auroralib/compression_amd/csrc/_obj
bash auroralib/compression_amd/csrc/build.sh > /dev/null 2>&1
