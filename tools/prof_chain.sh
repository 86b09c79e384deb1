# per-kernel times of the UNet forward (graph of 20 forwards x 11 replays at rows 128 / H 32 and rows 2 / H 16): bash tools/prof_chain.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_unet
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_unet -o unet -- python tools/chain_time.py > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/prof_unet/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
by = collections.defaultdict(list)
for r in rows:
    key = (r["Kernel_Name"].split("(")[0][-42:], r["Grid_Size_X"], r["LDS_Block_Size"])
    by[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in by.values())
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1]))[:40]:
    v.sort()
    print(f"{k[0]:42s} grid {k[1]:>7s} lds {k[2]:>6s}  n {len(v):5d}  median {v[len(v)//2]:7.2f} us  total {sum(v)/1e3:8.2f} ms ({100*sum(v)/tot:4.1f} %)")
PY
