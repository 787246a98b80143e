"""rocprofv3 --kernel-trace CSV -> the launches of the LAST forward in order: start offset (us), duration (us), grid, short kernel name."""
import csv
import glob
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
path = glob.glob(f"{src}/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
    n = n.replace("void ", "")
    return n[:110]


# a forward starts at the albert_embed kernel
starts = [i for i, r in enumerate(rows) if "albert_embed" in r["Kernel_Name"]]
first = starts[-1] - 3 if starts else 0
sel = rows[max(first, 0):]
t0 = int(sel[0]["Start_Timestamp"])
with open(dst, "w") as f:
    tot = 0.0
    for r in sel:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        tot += d
        grid = "x".join(str(int(r[k]) // max(1, int(r[w]))) for k, w in (("Grid_Size_X", "Workgroup_Size_X"), ("Grid_Size_Y", "Workgroup_Size_Y"), ("Grid_Size_Z", "Workgroup_Size_Z")) if k in r)
        f.write(f"{(int(r['Start_Timestamp']) - t0) / 1e3:10.1f} {d:9.1f} {grid:>16} {short(r['Kernel_Name'])}\n")
    f.write(f"# {len(sel)} launches, sum of durations {tot / 1e3:.3f} ms, span {(int(sel[-1]['End_Timestamp']) - t0) / 1e6:.3f} ms\n")
print(open(dst).read()[-400:])
