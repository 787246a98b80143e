# rocprofv3 kernel stats of the CSM-1B prompt block (190 positions, B = 8, bf16 weights): gpurun_out/<tag>/prefill_kernel_stats.txt
set -e
tag=${1:-csmprefill}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $root
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o k -- python3 tools/bench_csm.py --weights bfloat16 --frames 2 --prompt 190 > $out/prefill_bench.json 2> $out/kt.err
f=$(find $out/kt -name '*kernel_stats.csv' | head -1)
python3 - $f > $out/prefill_kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.2f} us  total {float(r['TotalDurationNs'])/1e6:8.3f} ms")
PY
cat $out/prefill_kernel_stats.txt
