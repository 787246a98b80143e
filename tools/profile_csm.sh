# rocprofv3 kernel stats of the CSM-1B frame bench (bf16 weights, B = 8, 20 single-token frames after the prompt):
#   bash tools/profile_csm.sh <tag>  ->  gpurun_out/<tag>/csm_kernel_stats.csv + csm_bench_under_rocprof.json
set -e
tag=${1:-csmprof}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $root
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o k -- python3 tools/bench_csm.py --weights bfloat16 --frames 20 > $out/csm_bench_under_rocprof.json 2> $out/kt.err
f=$(find $out/kt -name '*kernel_stats.csv' | head -1)
cp $f $out/csm_kernel_stats.csv
python3 - $out/csm_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print(f"{r['Name'][:90]:90s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:8.2f} us  {float(r['Percentage']):6.2f} %")
PY
