"""A/B of an environment switch on the deployed tick (B = 1, H = 16, 50 steps, one graph), same library, same GPU:
python tools/ab_env_tick.py ADX_PIPE_SENTINEL 0 1"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = "import sys; sys.path.insert(0, %r); import torch, bench, json; print(json.dumps(bench.deployed_leg(torch.device('cuda:0'))))" % root
name, vals = sys.argv[1], sys.argv[2:]
for rnd in range(2):
    for v in vals:
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **{name: v}), capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        print(rnd, name, v, json.loads(line[-1]).get("tick_ms_graph") if line else out.stderr[-300:])
