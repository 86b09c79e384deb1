"""Median kernel duration per layer from a rocprofv3 --kernel-trace CSV of tools/bench_tconv.py: the tool launches every
layer 3 + N times in a row, so consecutive chunks of the (time-ordered) tconv dispatches are the layers."""
import csv
import glob
import statistics
import sys

d, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 33
path = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(path)) if "tconv" in r["Kernel_Name"] and "pack" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tot = 0.0
for i in range(0, len(rows), n):
    ch = rows[i:i + n]
    du = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in ch[3:]]
    name = ch[0]["Kernel_Name"].split("(")[0].replace("void adx::", "")
    print(f"{i // n:2d} {name:34s} grid {ch[0].get('Grid_Size', '?'):>7s} wg {ch[0].get('Workgroup_Size', '?'):>5s} lds {ch[0].get('LDS_Block_Size', '?'):>7s} "
          f"vgpr {ch[0].get('VGPR_Count', '?'):>4s}: median {statistics.median(du):7.2f} us  min {min(du):7.2f}")
