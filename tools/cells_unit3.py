import sys, torch
sys.path.insert(0, ".")
from autonomous_driving_with_diffusion_model_amd import ops
DEV = "cuda:0"
torch.manual_seed(0)
for (cin, cout, n, h, w) in ((256, 256, 20, 16, 57), (512, 512, 40, 8, 29), (256, 256, 64, 16, 57)):
    x = torch.randn(n, cin, h, w, device=DEV)
    wt = torch.randn(cout, cin, 3, 3, device=DEV) * 0.05
    res = torch.randn(n, cout, h, w, device=DEV)
    y0, packed = ops.conv2d(x, wt, stride=1, pad=1, res=res, relu=True)
    xc, rc = ops.to_cells(x), ops.to_cells(res)
    print("nan in cells?", torch.isnan(ops.from_cells(xc, x.shape)).sum().item(), torch.isnan(ops.from_cells(rc, res.shape)).sum().item())
    for a, b in ((False, False), (True, False), (False, True), (True, True)):
        yc = ops.conv2d_cells(xc if a else x, packed, cin, cout, n, h, w, x_cells=a, res=rc if b else res, res_cells=b, relu=True)
        y = ops.from_cells(yc, y0.shape)
        bad = ~torch.isfinite(y)
        err = torch.where(bad, torch.zeros_like(y), (y - y0).abs())
        print((cin, cout, n, h, w), a, b, "nan", bad.sum().item(), "err", err.max().item(),
              "nan imgs", sorted(set(torch.nonzero(bad)[:, 0].tolist()))[:8], "chans", sorted(set(torch.nonzero(bad)[:, 1].tolist()))[:8])
