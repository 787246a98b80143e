# TIMING-ONLY sweep of the fused GEMV's phases (KK_CSM_DBG bits, wrong results): gpurun_out/<tag>/csm_dbg.txt
tag=${1:-csmdbg}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
for d in 0 1 2 4 8 3 6 7 15; do
  echo "== KK_CSM_DBG=$d" >> $out/csm_dbg.txt
  KK_CSM_DBG=$d python3 tools/bench_csm.py --weights bfloat16 --frames 20 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_frame'])" >> $out/csm_dbg.txt || exit 1
done
cat $out/csm_dbg.txt
