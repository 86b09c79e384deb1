cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ADX_CHAIN_DEBUG=1 ADX_CHAIN_MASK=0x3 python tools/chain_time.py 2>&1 | grep -E "chain\]|rows" | sort | uniq -c | head
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3c/prof_chain -o chain -- python tools/chain_time.py > /dev/null 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r3c/prof_chain/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:14]:
        print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:8.2f} us  total {float(r["TotalDurationNs"])/1e6:8.2f} ms')
PY
