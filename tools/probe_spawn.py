"""Does a process that has initialised the GPU may start a child python that uses the GPU too?  (test design probe)"""
import subprocess, sys, torch
print("parent cuda:", torch.cuda.is_available(), torch.zeros(1, device="cuda").item())
r = subprocess.run([sys.executable, "-c", "import torch; print('child cuda:', torch.cuda.is_available(), torch.ones(1, device='cuda').item())"],
                   capture_output=True, text=True, timeout=300)
print("child rc", r.returncode, r.stdout[-500:], r.stderr[-1500:])
