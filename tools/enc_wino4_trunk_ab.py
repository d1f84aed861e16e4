import sys, torch
sys.path.insert(0, '/root/repo')
import diinn_amd.modules as M
dev = torch.device("cuda:0")
enc = M.make_rdn().to(dev).eval()
def t_ms(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
with torch.no_grad():
    for lr in [int(a) for a in sys.argv[1:]] or [192, 256, 384]:
        x = torch.rand(1, 3, lr, lr, device=dev)
        enc.hip_winograd4 = False; a = enc(x); ta = t_ms(lambda: enc(x))
        enc.hip_winograd4 = True; b = enc(x); tb = t_ms(lambda: enc(x))
        enc.hip_trunk_max_pixels = None; r = enc(x); enc.hip_trunk_max_pixels = M.RDN.hip_trunk_max_pixels
        print(f"{lr}: F(2,3) {ta:.3f} ms  F(4,3) {tb:.3f} ms   |F23-miopen| {(a-r).abs().max().item():.2e} |F43-miopen| {(b-r).abs().max().item():.2e} max|ref| {r.abs().max().item():.3f}", flush=True)
