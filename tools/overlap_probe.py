"""Would the training backward gain from running the weight gradients beside the data gradients / BatchNorm passes?  Per layer
shape at B = 64: the weight gradient (A), and the chain [data-gradient-like 3x3 conv on the fp32 layout + an elementwise pass over
three tensors of the map's size] (B), timed one after the other on one stream and side by side on two streams."""
import sys
import torch
sys.path.insert(0, ".")
from autonomous_driving_with_diffusion_model_amd import ops

dev = torch.device("cuda:0")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(c, h, w, iters=30):
    g = torch.Generator().manual_seed(c)
    x = torch.randn(64, c, h, w, generator=g).to(dev)
    dy = (torch.randn(64, c, h, w, generator=g) * 1e-3).to(dev)
    wt = (torch.randn(c, c, 3, 3, generator=g) * 0.05).to(dev)
    y, packed = ops.conv2d(dy, wt, pad=1)
    t1, t2 = torch.empty_like(x), torch.empty_like(x)

    def A():
        ops.conv2d_weight_grad(x, dy, 3, pad=1)

    def B():
        ops.conv2d(dy, wt, pad=1, packed=packed, out=y)
        torch.add(y, x, out=t1)             # stand-in for the BatchNorm-backward apply pass: two reads, one write
        torch.mul(t1, 0.5, out=t2)

    def timed(fn):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    def both_seq():
        A(); B()

    def both_par():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            A()
        with torch.cuda.stream(s2):
            B()
        cur.wait_stream(s1); cur.wait_stream(s2)

    ta, tb, ts, tp = timed(A), timed(B), timed(both_seq), timed(both_par)
    print(f"{c:4d} ch @{h}x{w}: wgrad {ta:.3f}  conv+passes {tb:.3f}  one stream {ts:.3f}  two streams {tp:.3f} ms  ({100 * (1 - tp / ts):+.1f} %)")


for c, h, w in ((64, 64, 225), (128, 32, 113), (256, 16, 57), (512, 8, 29)):
    run(c, h, w)
    run(c, h, w)
