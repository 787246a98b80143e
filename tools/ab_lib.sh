# A/B of a side build (mlx-audio_amd/libkokoro_hip_<name>.so, e.g. build.py --mfma32) against the product library, alternating on ONE box:
#   bash tools/ab_lib.sh <tag> <name> [shapes...]   ->  gpurun_out/<tag>/ab_<name>.txt
tag=$1; name=$2; shift 2
shapes=${@:-st1_k3 st1_k7_d3 st1_k11_d5 st0_k3 st0_k7 st0_k11_d5 dec_1024_1024_k3 dec_1152_1024_k3 albert_qkv albert_ffn}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
for rep in 1 2; do
for lib in product $name; do
  if [ $lib = product ]; then unset KK_HIP_LIB; else export KK_HIP_LIB=$root/mlx-audio_amd/libkokoro_hip_$name.so; fi
  for mode in "--fused --v4" "--fused --v5" "--v4"; do
    echo "== $lib $mode (rep $rep)" >> $out/ab_$name.txt
    python3 tools/bench_conv.py $mode $shapes 2>/dev/null >> $out/ab_$name.txt || exit 1
  done
done
done
python3 - <<PY
import json,collections
rows=collections.defaultdict(lambda: collections.defaultdict(list)); cur=None
for l in open("$out/ab_$name.txt"):
    if l.startswith("=="): cur=l.split()[1]; continue
    if l.startswith("{"):
        d=json.loads(l); rows[d["name"]][cur].append(d["ms"])
for k,v in rows.items():
    a=min(v["product"]); b=min(v["$name"]); print(f"{k:34s} product {a:.4f} ms   $name {b:.4f} ms   product/{'$name'} = {a/b:.3f}")
PY
