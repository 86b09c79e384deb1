"""Idle time of the GPU inside a training step, from a `rocprofv3 --kernel-trace --output-format csv` run of tools/train_time.py.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr_trace -- python3 tools/train_time.py
    python tools/train_gaps.py gpurun_out/tr_trace

A step ends with `adamw_ema_kernel`.  For the last steps of the run: span (first start .. last end), the union of the kernels'
busy intervals, and the idle gaps between one kernel's end and the next one's start, summed per (kernel before, kernel after).
"""
import csv
import glob
import sys
from collections import defaultdict


def short(n):
    n = n.replace("void ", "").replace("adx::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:44]


def main(d):
    f = sorted(glob.glob(d + "/*/*kernel_trace.csv"))[-1]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
    rows.sort()
    steps, cur = [], []
    for r in rows:
        cur.append(r)
        if "adamw_ema_kernel" in r[2]:
            steps.append(cur)
            cur = []
    steps = steps[len(steps) // 2:]           # the first ones build workspaces
    tot_span = tot_busy = 0.0
    gaps = defaultdict(lambda: [0, 0.0])
    phase_gap = defaultdict(float)
    for st in steps:
        span = st[-1][1] - st[0][0]
        busy, end = 0, st[0][0]
        for i, (s, e, n) in enumerate(st):
            if s > end:
                g = s - end
                if g > 1000:
                    k = (short(st[i - 1][2]), short(n))
                    gaps[k][0] += 1
                    gaps[k][1] += g
            if e > end:
                busy += e - max(s, end)
                end = e
        tot_span += span
        tot_busy += busy
    n = len(steps)
    print(f"{n} steps: span {tot_span / n / 1e6:.3f} ms per step, busy {tot_busy / n / 1e6:.3f} ms, idle {(tot_span - tot_busy) / n / 1e6:.3f} ms")
    print("idle gaps > 1 us, per step, by (kernel before -> kernel after):")
    for k, (c, g) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"  {g / n / 1e3:8.1f} us  x{c / n:5.1f}  {k[0]} -> {k[1]}")
    # the neighbourhood of the last step's largest gap (and any memory copies the run traced there)
    st = steps[-1]
    gi = max(range(1, len(st)), key=lambda i: st[i][0] - max(x[1] for x in st[max(0, i - 8):i]))
    t0 = st[gi - 1][1]
    print("around the largest gap of the last step (us relative to the end of the kernel before it):")
    for s_, e_, n_ in st[max(0, gi - 6):gi + 6]:
        print(f"  {(s_ - t0) / 1e3:9.1f} .. {(e_ - t0) / 1e3:9.1f}  {short(n_)}")
    for mf in glob.glob(d + "/*/*memory_copy_trace.csv"):
        for r in csv.DictReader(open(mf)):
            s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            if st[max(0, gi - 6)][0] <= s_ <= st[min(len(st) - 1, gi + 5)][1]:
                print(f"  {(s_ - t0) / 1e3:9.1f} .. {(e_ - t0) / 1e3:9.1f}  memory copy {r.get('Direction', '')} {r.get('Bytes', '')}")
    # time between steps (host: optimizer bookkeeping, next batch)
    if len(steps) > 1:
        between = [steps[i + 1][0][0] - steps[i][-1][1] for i in range(len(steps) - 1)]
        print(f"between steps (last kernel end -> next step's first kernel start): {sum(between) / len(between) / 1e3:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1])
