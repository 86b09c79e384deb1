"""A/B two builds of libadx on the SAME GPU: the deployed tick (B = 1, H = 16, 50 steps, one graph).  ADX_LIB selects the library."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = "import sys; sys.path.insert(0, %r); import torch, bench, json; print(json.dumps(bench.deployed_leg(torch.device('cuda:0'))))" % root
for rnd in range(2):
    for name in sys.argv[1:]:
        env = dict(os.environ, ADX_LIB=os.path.join(root, name))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        print(rnd, name, json.loads(line[-1]).get("tick_ms_graph") if line else out.stderr[-300:])
