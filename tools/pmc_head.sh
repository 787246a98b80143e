# SQ counters of the fused conv_post + iSTFT head kernel (what bounds it): separate --pmc passes over tools/bench_head.py
#   bash tools/pmc_head.sh <tag>  ->  gpurun_out/<tag>/pmc_head_*.csv  (digest: tools/pmc_head_digest.py)
tag=${1:-pmch}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS"; do
  t=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$t -o q -- python3 tools/bench_head.py > $out/pmc_$t.log 2>&1 || echo "fail $t"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("$out/pmc_*/q_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "conv_post_istft" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, v) in sorted(agg.items()):
    print(f"{k:28s} launches {n:3d}  per launch {v / max(n, 1):.4g}")
PY
