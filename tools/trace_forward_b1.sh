# Per-launch kernel trace of ONE eager forward at B = 1 (the p50-latency configuration): gpurun_out/<tag>/forward_trace_b1.txt + top kernels by time
set -e
tag=${1:-traceb1}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $root
rocprofv3 --kernel-trace --output-format csv -d $out/kt -o k -- python3 bench.py --batch 1 --streams 1 --steps 1 --warmup 1 --no-graph --no-profile --no-latency --no-cpu-baseline > $out/trace_bench.json 2> $out/kt.err
python3 tools/trace_digest.py $out/kt $out/forward_trace_b1.txt
rm -f $out/kt/*kernel_trace.csv
python3 - $out/forward_trace_b1.txt <<'PY'
import re, sys, collections
rows = []
for l in open(sys.argv[1]):
    m = re.match(r'\s*([\d.]+)\s+([\d.]+)\s+(\S+)\s+(.*)', l)
    if m: rows.append((float(m.group(1)), float(m.group(2)), m.group(3), m.group(4)))
print('launches', len(rows), 'sum of durations us', round(sum(r[1] for r in rows), 1), 'span us', round(rows[-1][0] + rows[-1][1] - rows[0][0], 1))
agg = collections.defaultdict(lambda: [0, 0.0])
for t, d, g, n in rows:
    k = n.split('(')[0][:70]
    agg[k][0] += 1; agg[k][1] += d
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"{v[1]:9.1f} us {v[0]:4d}x {v[1]/v[0]:8.1f}  {k}")
PY
