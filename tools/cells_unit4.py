import sys, torch
sys.path.insert(0, ".")
from autonomous_driving_with_diffusion_model_amd import ops
DEV = "cuda:0"
torch.manual_seed(0)
cin, cout, n, h, w = 256, 256, 20, 16, 57
x = torch.randn(n, cin, h, w, device=DEV)
wt = torch.randn(cout, cin, 3, 3, device=DEV) * 0.05
y0, packed = ops.conv2d(x, wt, stride=1, pad=1)
for rep in range(2):
    yc = ops.conv2d_cells(x, packed, cin, cout, n, h, w, x_cells=False)
    y = ops.from_cells(yc, y0.shape)
    bad = ((y - y0).abs() > 1e-3) | ~torch.isfinite(y)
    idx = torch.nonzero(bad)
    print("rep", rep, "bad", idx.shape[0], "of", y.numel())
    print(" imgs", sorted(set(idx[:, 0].tolist()))[:10], " chan%64", sorted(set((idx[:, 1] % 64).tolist())), " rows", sorted(set(idx[:, 2].tolist())), " cols", sorted(set(idx[:, 3].tolist()))[:40])
    # raw view: which plane is bad? compare hi/lo halves
    v = yc.view(torch.float16).view(n, cout // 8, 2, h, w, 8).float()
    ref = ops.to_cells(y0).view(torch.float16).view(n, cout // 8, 2, h, w, 8).float()
    for pl in (0, 1):
        d = (v[:, :, pl] != ref[:, :, pl])
        print("  plane", pl, "mismatching halves", d.sum().item())
