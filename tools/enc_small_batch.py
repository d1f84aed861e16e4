"""r04: the HIP encoder trunk on small maps, B = 1 vs B = 2 / 4 (is the chip idle at 48x48?)  python tools/enc_small_batch.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import diinn_amd.modules as M
dev = torch.device("cuda:0")
torch.manual_seed(0)
enc = M.make_rdn().to(dev).eval()
for size in (48, 64, 96):
    for b in (1, 2, 4):
        x = torch.rand(b, 3, size, size, device=dev)
        with torch.no_grad():
            for _ in range(3): enc(x)
            torch.cuda.synchronize()
            ts = []
            for _ in range(10):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); enc(x); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        ts.sort()
        print(f"{size}x{size} B={b}: {ts[len(ts)//2]:.3f} ms per forward, {ts[len(ts)//2]/b:.3f} ms per image")
