"""CPU simulation of Winograd F(2x2,3x3) / split-fp16 numerics against fp64: relative rms
error of the direct split product, plain fp32, and the Winograd forms on one K = 4608 case."""
import torch, numpy as np
torch.manual_seed(0)
C, K, H, W = 512, 64, 8, 30   # Cin, Cout
x = torch.randn(1, C, H + 2, W + 2).clamp_min(0) * 1.0   # post-ReLU like
x = torch.randn(1, C, H + 2, W + 2)
w = torch.randn(K, C, 3, 3) * (2.0 / (9 * C)) ** 0.5
ref = torch.nn.functional.conv2d(x.double(), w.double())
def split(t):
    h = t.half().float(); l = ((t - h) * 2048).half().float(); return h, l
def mm3(a, b):   # a [.., M, K] b [.., K, N]  split product, fp32 accumulate
    ah, al = split(a); bh, bl = split(b)
    return ah @ bh + (ah @ bl + al @ bh) * (1.0 / 2048)
# direct (im2col)
cols = torch.nn.functional.unfold(x, 3)[0]            # [C*9, L]
d = mm3(w.reshape(K, -1), cols).reshape(1, K, H, W)
d32 = (w.reshape(K, -1) @ cols).reshape(1, K, H, W)
def err(y): return ((y.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
print("direct split", err(d), "direct fp32", err(d32))
# winograd F(2x2,3x3)
Bt = torch.tensor([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]], dtype=torch.float32)
G = torch.tensor([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], dtype=torch.float64)
At = torch.tensor([[1,1,1,0],[0,1,-1,-1]], dtype=torch.float32)
U = (G @ w.double() @ G.T).float()                     # [K,C,4,4]
tiles = x.unfold(2, 4, 2).unfold(3, 4, 2)              # [1,C,th,tw,4,4]
V = Bt @ tiles @ Bt.T                                  # fp32
th, tw = V.shape[2], V.shape[3]
Vp = V.permute(4, 5, 1, 0, 2, 3).reshape(4, 4, C, th * tw)
Up = U.permute(2, 3, 0, 1)                             # [4,4,K,C]
for name, f in (("wino split", mm3), ("wino fp32", lambda a, b: a @ b), ("wino fp64 products", lambda a, b: (a.double() @ b.double()).float())):
    M = f(Up, Vp)                                      # [4,4,K,T]
    M = M.permute(2, 3, 0, 1).reshape(K, th, tw, 4, 4)
    Y = At @ M @ At.T                                  # [K,th,tw,2,2]
    y = Y.permute(0, 1, 3, 2, 4).reshape(1, K, th * 2, tw * 2)
    print(name, err(y))
