cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmcq_$tag -o q -- python3 tools/bench_conv.py --fused --v4 st1_k3 st1_k11_d5 st0_k7 > gpurun_out/pmcq_$tag.log 2>&1 || echo "fail $tag"
done
ls gpurun_out | grep pmcq
