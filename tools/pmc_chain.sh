# SQ counters of the chained-level kernel (one UNet forward at rows 2, H 16, eager): bash tools/pmc_chain.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/one_fwd.py <<'PY'
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from autonomous_driving_with_diffusion_model_amd.utils import procedural as P
from test_gpu_model import make_model
import os
rows, H = int(os.environ.get("ROWS", "2")), int(os.environ.get("H", "16"))
m, _ = make_model("FREE_GUIDANCE", H)
d = P.synthetic_batch(rows, H, image_hw=(32, 32), seed=12)
feat = P._uniform("feat", 12, (rows, 64), -3.0, 3.0).to("cuda:0")
m.perception.forward = lambda img: feat
with torch.no_grad():
    for _ in range(4):
        m(d["trajs"].to("cuda:0"), d["imgs"].to("cuda:0"), d["t"].to("cuda:0"), cond=d["target"].to("cuda:0"))
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d gpurun_out/pmc_chainA -- python3 /tmp/one_fwd.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_IFETCH SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_WAVES --output-format csv -d gpurun_out/pmc_chainB -- python3 /tmp/one_fwd.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_INSTS_FLAT SQ_INSTS_WAVE32_LDS GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_chainC -- python3 /tmp/one_fwd.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
for d in sorted(glob.glob("gpurun_out/pmc_chain[ABC]")):
    f = glob.glob(d + "/*/*counter_collection.csv") + glob.glob(d + "/*counter_collection.csv")
    if not f: print(d, "no csv"); continue
    acc = {}
    for r in csv.DictReader(open(f[0])):
        for key in ("tconv_chain_kernel", "tconv_hsd_kernel", "tconv_hs_kernel<2", "tconv_hs_kernel<1"):
            if key in r["Kernel_Name"]:
                acc.setdefault((key, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(d.split("/")[-1], k, "launches", len(v), "mean per launch %.0f" % (sum(v) / len(v)))
PY
