import sys, torch
sys.path.insert(0, ".")
from autonomous_driving_with_diffusion_model_amd import ops
DEV = "cuda:0"
torch.manual_seed(0)
cin, cout, n, h, w = 64, 64, 2, 64, 225
x = torch.randn(n, cin, h, w, device=DEV)
wt = torch.randn(cout, cin, 3, 3, device=DEV) * 0.05
y0, packed = ops.conv2d(x, wt, stride=1, pad=1)
yc = ops.conv2d_cells(x, packed, cin, cout, n, h, w, x_cells=False)
y = ops.from_cells(yc, y0.shape)
err = (y - y0).abs()
print("per-channel max err:", [round(v, 3) for v in err.amax(dim=(0, 2, 3)).tolist()])
# does y channel c equal y0 channel c' for some c'?
for c in range(0, 16):
    d = [(y[:, c] - y0[:, c2]).abs().max().item() for c2 in range(64)]
    print(c, "best match", min(range(64), key=lambda i: d[i]), min(d))
