#!/usr/bin/env python3
"""What the memory system takes for the q / k / v projection's traffic pattern, without any arithmetic: read R MB, write W MB with plain torch kernels
(fill, copy, cat) at the C2 self-launch size (67 MB of x rows in, 201 MB of planes out) -- the roof the projection kernels are compared with."""
import torch
dev = torch.device("cuda:0")
def timed(fn, reps=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
MB = 1 << 20
x = torch.randn(64 * MB // 4, device=dev)                    # 64 MiB
y3 = torch.empty(3 * x.numel(), device=dev)                  # 192 MiB
y1 = torch.empty_like(x)
big = torch.randn(256 * MB // 4, device=dev)
for name, fn, nbytes in (
    ("fill 192 MiB (write only)", lambda: y3.fill_(1.0), y3.numel() * 4),
    ("copy 64 -> 64 MiB", lambda: y1.copy_(x), 2 * x.numel() * 4),
    ("read 64, write 192 MiB (cat of three)", lambda: torch.cat([x, x, x], out=y3), 4 * x.numel() * 4),
    ("read 256 MiB (sum)", lambda: big.sum(), big.numel() * 4),
    ("copy 256 -> 256 MiB", lambda: big.clone(), 2 * big.numel() * 4),
):
    us = timed(fn)
    print(f"{name:42s} {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s", flush=True)
