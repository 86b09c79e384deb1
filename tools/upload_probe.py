"""What handing the path HOST buffers would add: host -> device time of the tick's image (the reference's callers build it on the host:
interact.py:118-128, diffusion_agent.py:140-160), pageable and pinned, as u8 HWC frames (adx_resnet_forward_u8's input) and as fp32 NCHW."""
import time
import torch
dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
for name, shape, dt in (("B=1 u8 HWC", (1, 256, 900, 3), torch.uint8), ("B=1 f32 NCHW", (1, 3, 256, 900), torch.float32),
                        ("B=64 u8 HWC", (64, 256, 900, 3), torch.uint8), ("B=64 f32 NCHW", (64, 3, 256, 900), torch.float32)):
    for pinned in (False, True):
        h = torch.zeros(shape, dtype=dt)
        if pinned:
            h = h.pin_memory()
        d = torch.empty(shape, dtype=dt, device=dev)
        for _ in range(3):
            d.copy_(h, non_blocking=pinned)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            d.copy_(h, non_blocking=pinned)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print(f"{name:14s} {'pinned  ' if pinned else 'pageable'} {h.numel() * h.element_size() / 1e6:8.2f} MB  {ms:8.3f} ms  {h.numel() * h.element_size() / ms / 1e6:6.1f} GB/s")
