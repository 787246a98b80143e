# rocprofv3 kernel stats of config 4 with four jobs in flight (bench.py --config csm): where the GPU time of a job goes
set -e
tag=${1:-csmjobs}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $root
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o k -- python3 bench.py --config csm --steps 4 --warmup 1 --no-cpu-baseline > $out/bench.json 2> $out/kt.err
f=$(find $out/kt -name '*kernel_stats.csv' | head -1)
python3 - $f > $out/kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:26]:
    print(f"{r['Name'][:96]:96s} {int(r['Calls']):7d} {float(r['AverageNs'])/1e3:9.2f} us  {float(r['TotalDurationNs'])/tot*100:6.2f} %")
PY
cat $out/kernel_stats.txt
