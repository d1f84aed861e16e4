"""The whole RDN encoder with its 3x3 layers on the split-bf16 kernel (RDN.hip_split_bf16 = True) against the fp32 Winograd
trunk: time, feature difference, and the difference of the decoded image.  usage: python tools/enc_x3_time.py [SIZE ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import diinn_amd.modules as M

def t_ms(fn, n=5):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = M.DIINN(mode=3, init_q=False).to(dev).eval()
enc = net.encoder
sizes = [int(a) for a in sys.argv[1:]] or [224, 256, 384, 512]
with torch.no_grad():
    for lr in sizes:
        x = torch.rand(1, 3, lr, lr, device=dev)
        enc.hip_split_bf16 = False
        f32 = enc(x); t32 = t_ms(lambda: enc(x))
        img32 = net.decoder(f32, (2 * lr, 2 * lr), 30000)
        enc.hip_split_bf16 = True
        f3 = enc(x); t3 = t_ms(lambda: enc(x))
        img3 = net.decoder(f3, (2 * lr, 2 * lr), 30000)
        enc.hip_split_bf16 = False
        print(f"LR {lr}x{lr}: encoder fp32 Winograd {t32:.2f} ms | split-bf16 3x3 layers {t3:.2f} ms; features max|f| {float(f32.abs().max()):.2f} "
              f"differ by {float((f3 - f32).abs().max()):.2e}; decoded x2 image max {float(img32.abs().max()):.3f} differs by {float((img3 - img32).abs().max()):.2e}", flush=True)
