"""End-to-end DIINN forward (PyTorch-ROCm RDN encoder + HIP decoder) timing: where the time goes
once the decoder is fast.  Not the bench metric."""
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
net = M.DIINN(mode=3, init_q=False).to(dev).eval()
for tag in ("default", "miopen-benchmark", "channels_last+benchmark"):
    if tag != "default":
        torch.backends.cudnn.benchmark = True
    enc = net.encoder
    if tag.startswith("channels_last"):
        enc = enc.to(memory_format=torch.channels_last)
    for lr, s in ((48, 2), (256, 4)):
        x = torch.rand(1, 3, lr, lr, device=dev)
        if tag.startswith("channels_last"):
            x = x.contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            feat = enc(x)
            te = t_ms(lambda: enc(x))
            td = t_ms(lambda: net.decoder(feat, (lr * s, lr * s), 30000))
        print(f"[{tag}] LR {lr}x{lr} x{s}: encoder {te:.2f} ms ({43.9e6*lr*lr/te/1e9:.1f} TFLOP/s), decoder {td:.2f} ms", flush=True)

net.graphs = True
torch.backends.cudnn.benchmark = False
for lr, s in ((48, 2), (128, 4), (256, 4)):
    x = torch.rand(1, 3, lr, lr, device=dev)
    with torch.no_grad():
        net.graphs = False
        te = t_ms(lambda: net(x, (lr * s, lr * s), 30000))
        net.graphs = True
        tg = t_ms(lambda: net(x, (lr * s, lr * s), 30000))
    print(f"[end-to-end] LR {lr}x{lr} x{s}: eager {te:.2f} ms, hipGraph {tg:.2f} ms", flush=True)
