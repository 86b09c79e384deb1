export HIP_FORCE_DEV_KERNARG=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tr -- python3 tools/train_time.py > gpurun_out/prof_tr.log 2>&1
f=$(ls -t gpurun_out/prof_tr/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r['TotalDurationNs']) for r in rows)
print("total kernel ms per step", tot / 8e6)
for r in rows[:26]:
    print(f"{r['Name'][:66]:66s} calls/step {int(r['Calls'])/8:6.1f} ms/step {int(r['TotalDurationNs'])/8e6:6.3f} avg {float(r['AverageNs'])/1e3:7.1f} us")
PY
