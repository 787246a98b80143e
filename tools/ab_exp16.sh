# A/B of the TIMING-ONLY 16x16x32 build (build.py --exp16) against the product library on one box: gpurun_out/<tag>/ab_exp16.txt
tag=${1:-exp16}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
for rep in 1 2; do
for lib in base exp16; do
  if [ $lib = exp16 ]; then export KK_HIP_LIB=$root/mlx-audio_amd/libkokoro_hip_exp16.so; else unset KK_HIP_LIB; fi
  for mode in "--fused --v4" "--fused --v5" "--v4"; do
    echo "== $lib $mode (rep $rep)" >> $out/ab_exp16.txt
    python3 tools/bench_conv.py $mode st1_k3 st1_k7_d3 st1_k11_d5 st0_k3 st0_k7 st0_k11_d5 dec_1024_1024_k3 >> $out/ab_exp16.txt 2>&1 || exit 1
  done
done
done
cat $out/ab_exp16.txt
