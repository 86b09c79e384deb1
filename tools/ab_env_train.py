"""A/B of an environment switch on the training step, same library, same GPU: python tools/ab_env_train.py ADX_BWD_STREAMS 0 1"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = "import sys; sys.path.insert(0, %r); import torch, bench, json; print(json.dumps(bench.train_leg(torch.device('cuda:0'), 1, steps=6, warm=2)))" % root
name, vals = sys.argv[1], sys.argv[2:]
for rnd in range(2):
    for v in vals:
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **{name: v}), capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        print(rnd, name, v, json.loads(line[-1])["ms_per_step"] if line else out.stderr[-300:])
