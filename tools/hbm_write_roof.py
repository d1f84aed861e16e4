"""What does a pure store stream reach on this box?  (the bf16 P kernel writes 3.69 GB of fp32 P at c5 in 1.29 ms)
usage: python tools/hbm_write_roof.py"""
import torch
dev = torch.device("cuda:0")
n = 720 * 1280 * 1024
x = torch.empty(n, device=dev)
y = torch.randn(n // 4, device=dev)
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: x.zero_())
print(f"zero_  {n * 4 / 1e9:.2f} GB: {ms:.3f} ms = {n * 4 / ms / 1e9:.2f} TB/s")
ms = t(lambda: x.fill_(1.5))
print(f"fill_  {n * 4 / 1e9:.2f} GB: {ms:.3f} ms = {n * 4 / ms / 1e9:.2f} TB/s")
z = torch.empty(n // 4, device=dev)
ms = t(lambda: z.copy_(y))
print(f"copy_  {n / 1e9:.2f} GB read + {n / 1e9:.2f} GB write: {ms:.3f} ms = {2 * n / ms / 1e9:.2f} TB/s")
ms = t(lambda: torch.mul(y, 2.0, out=z))
print(f"mul    {n / 1e9:.2f} GB read + {n / 1e9:.2f} GB write: {ms:.3f} ms = {2 * n / ms / 1e9:.2f} TB/s")
