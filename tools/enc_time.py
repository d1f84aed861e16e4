import sys
sys.path.insert(0, "/root/repo")
import torch
import diinn_amd.modules as M
dev = torch.device("cuda:0")
enc = M.make_rdn().to(dev).eval()
def t_ms(fn, n=5):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for lr in (48, 256, 512):
    x = torch.rand(1, 3, lr, lr, device=dev)
    with torch.no_grad():
        a = enc(x)
        t1 = t_ms(lambda: enc(x))
    with torch.enable_grad():
        for p in enc.parameters(): p.requires_grad_(False)
        b = enc(x)
        t0 = t_ms(lambda: enc(x))
    print(f"LR {lr}: cat form {t0:.2f} ms, dense-buffer form {t1:.2f} ms, max diff {(a-b).abs().max().item():.2e}")
