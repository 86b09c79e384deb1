"""One-line summary of a bench.py JSON line: python tools/bench_summary.py gpurun_out/r05/bench.json"""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = d["roofline"]
print("faithful", d["value"], "steps/s", d["ms_per_step"], "ms | frac", r["frac"], "alone", r.get("frac_alone"), "launch", r["avg_launch_ms"], "ms")
print("per shape", [(p["shape"], p["ms"], p.get("ms_launched_alone")) for p in r.get("per_shape", [])])
print("sustained", r.get("sustained", {}).get("mfma_fp16_tflops_random_operands"), "issued/sust", r.get("sustained", {}).get("achieved_issued_over_sustained"))
print("hoisted", d["hoisted"]["value"], "graph", d["hoisted_graph"]["value"], "| train", d["train"]["ms_per_step"], "train_free", d["train_free"]["ms_per_step"],
      "| tick", d["deployed_b1_h16"]["tick_ms_graph"], "cls", d["deployed_b1_h16"]["classifier_guidance_tick_ms_graph"], "| tconv us", d["roofline_tconv"]["avg_launch_ms"] * 1e3,
      "| cpu", d["cpu_baseline"]["value"])
