# Round profiles on the GPU box (run through gpurun from the repo root): kernel-trace stats of the default bench command, then the two PMC
# passes (FETCH_SIZE / WRITE_SIZE never share a pass: MI355X_MICROARCH.md, TCC counter budget) over a 2-forward eager run.
# usage: bash tools/profile_round.sh <tag>   -> gpurun_out/<tag>/...
set -e
tag=${1:-prof}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $root
# (--streams 1: one batch in flight, so that a kernel's average duration in the summary is its stand-alone duration, as in bench.py's own brackets)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o k -- python3 bench.py --streams 1 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/kt.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -o q -- python3 bench.py --streams 1 --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-latency --no-graph > $out/pmc_$c.log 2>&1
done
python3 bench.py > $out/bench.json 2> $out/bench.err
ls -R $out | head -40
