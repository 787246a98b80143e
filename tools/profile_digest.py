"""Turns the output of tools/profile_round.sh (gpurun_out/<tag>/) into the committed round summaries under profiles/:
  python tools/profile_digest.py gpurun_out/r02p3 r02
  -> profiles/<round>_a_bf16_kernel_stats.csv            rocprofv3 --kernel-trace --stats of `bench.py --streams 1 ...`
     profiles/<round>_a_bf16_bench_under_rocprof.json    that run's own JSON line
     profiles/<round>_c_bf16_bench_two_in_flight.json    the default `bench.py` line
     profiles/<round>_pmc_traffic.json                   FETCH_SIZE / WRITE_SIZE per launch of the conv family and the fused head (bench.py reads it)
FETCH_SIZE / WRITE_SIZE come from two separate --pmc passes (they do not fit one pass, MI355X_MICROARCH.md) over 2 eager forwards; values are KiB."""
import collections
import csv
import json
import os
import shutil
import sys

src, rnd = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(ROOT, "profiles")
shutil.copy(f"{src}/kt/k_kernel_stats.csv", f"{prof}/{rnd}_a_bf16_kernel_stats.csv")
shutil.copy(f"{src}/bench_under_rocprof.json", f"{prof}/{rnd}_a_bf16_bench_under_rocprof.json")
shutil.copy(f"{src}/bench.json", f"{prof}/{rnd}_c_bf16_bench_two_in_flight.json")


def family(name: str):
    if "conv_mfma5_kernel" in name or "conv_mfma4_kernel" in name or "conv_mfma_kernel" in name:
        return "conv_mfma"  # variants 5 / 4 / 2 of the MFMA convolution (bench.py's conv_mfma bracket covers all three)
    if "conv_post_istft" in name:
        return "istft_head"
    return None


agg = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    a = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f"{src}/pmc_{c}/q_counter_collection.csv")):
        k = family(r["Kernel_Name"])
        if k:
            a[k][0] += 1
            a[k][1] += float(r["Counter_Value"]) * 1024.0
    agg[c] = a
nfwd = 2
path = f"{prof}/{rnd}_pmc_traffic.json"
d = json.load(open(path)) if os.path.exists(path) else {"bfloat16": {"istft_head": {}}}
d["bfloat16"]["conv_mfma"] = {"launches_per_step": agg["FETCH_SIZE"]["conv_mfma"][0] // nfwd,
                              "fetch_raw_bytes_per_step": agg["FETCH_SIZE"]["conv_mfma"][1] / nfwd,
                              "write_bytes_per_step": agg["WRITE_SIZE"]["conv_mfma"][1] / nfwd}
d["bfloat16"].setdefault("istft_head", {}).update({
    "kernel": "conv_post_istft_kernel (conv_post + iSTFT + overlap-add in one launch)", "launches_per_step": 1,
    "fetch_raw_bytes_per_launch": agg["FETCH_SIZE"]["istft_head"][1] / nfwd,
    "fetch_corrected_bytes_per_launch": 2 * agg["FETCH_SIZE"]["istft_head"][1] / nfwd,
    "write_bytes_per_launch": agg["WRITE_SIZE"]["istft_head"][1] / nfwd})
json.dump(d, open(path, "w"), indent=1)
print(json.dumps(d["bfloat16"], indent=1))
