# Per-launch kernel trace of ONE eager forward of the benchmark configuration (B = 32, T = 130, F = 650), in launch order:
#   bash tools/trace_forward.sh <tag>  ->  gpurun_out/<tag>/forward_trace.txt  (name, grid, duration; the last forward of the run)
set -e
tag=${1:-trace}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $root
rocprofv3 --kernel-trace --output-format csv -d $out/kt -o k -- python3 bench.py --streams 1 --steps 1 --warmup 1 --no-graph --no-profile --no-latency --no-cpu-baseline > $out/trace_bench.json 2> $out/kt.err
python3 tools/trace_digest.py $out/kt $out/forward_trace.txt
