export HIP_FORCE_DEV_KERNARG=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for shape in "512 8 29" "256 16 57" "64 64 225"; do
tag=$(echo $shape | tr ' ' '_')
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d gpurun_out/pmcA_$tag -- python3 tools/pmc_one.py $shape > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_ANY --output-format csv -d gpurun_out/pmcB_$tag -- python3 tools/pmc_one.py $shape > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob
res = {}
for d in sorted(glob.glob("gpurun_out/pmc[AB]_*")):
    f = glob.glob(d + "/*/*counter_collection.csv")
    if not f: print(d, "no csv"); continue
    acc = {}
    for r in csv.DictReader(open(f[0])):
        if "conv2d_hs3x3" in r["Kernel_Name"]:          # conv2d_hs3x3_kernel<...> (64 channels) and conv2d_hs3x3q_kernel<...>
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    res[d.split("/")[-1]] = {k: round(sum(v[1:]) / max(1, len(v) - 1)) for k, v in acc.items()}
    print(d, res[d.split("/")[-1]])
import json
json.dump({"source": "bash tools/pmc_sq.sh: rocprofv3 --kernel-trace --pmc <8 counters> -- python3 tools/pmc_one.py <cin> <h> <w> (B = 64, one 3x3 stride-1 conv with BN + residual + ReLU, per launch, averaged over 4 launches); pmcA / pmcB = the two counter sets", "kernel": "conv2d_hs3x3q_kernel<true> (512, 256 channels), conv2d_hs3x3_kernel<0, 0, true, true> (64 channels)", "counters": res}, open("gpurun_out/" + __import__("os").environ.get("R", "r02") + "_sq_counters.json", "w"), indent=1)
PY
