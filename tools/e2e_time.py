"""End-to-end DIINN forward timing: where the time goes.  Not the bench metric.
Encoder on the HIP trunk (conv_ksplit_kernel) vs on PyTorch-ROCm/MIOpen, decoder on the HIP path, then the
whole model eager and hipGraph-replayed."""
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
enc = net.encoder
default_cap = enc.hip_trunk_max_pixels
for lr, s in ((48, 2), (128, 4), (256, 4), (512, 4)):
    x = torch.rand(1, 3, lr, lr, device=dev)
    with torch.no_grad():
        enc.hip_trunk_max_pixels = None
        t_mi = t_ms(lambda: enc(x))
        enc.hip_trunk_max_pixels = default_cap
        feat = enc(x)
        t_hip = t_ms(lambda: enc(x))
        td = t_ms(lambda: net.decoder(feat, (lr * s, lr * s), 30000))
        net.graphs = False
        te = t_ms(lambda: net(x, (lr * s, lr * s), 30000))
        net.graphs = True
        tg = t_ms(lambda: net(x, (lr * s, lr * s), 30000))
        net.graphs = False
        net.decoder.compute = "bf16x3"                       # the optional split-bf16 decoder (fp32 tolerance)
        td3 = t_ms(lambda: net.decoder(feat, (lr * s, lr * s), 30000))
        te3 = t_ms(lambda: net(x, (lr * s, lr * s), 30000))
        enc.hip_split_bf16 = True                            # ... and the encoder's 3x3 layers in split bf16 as well
        t_hip3 = t_ms(lambda: enc(x))
        te33 = t_ms(lambda: net(x, (lr * s, lr * s), 30000))
        enc.hip_split_bf16 = False
        net.decoder.compute = "f32"
    print(f"LR {lr}x{lr} x{s}: encoder HIP trunk {t_hip:.2f} ms ({43.9e6*lr*lr/t_hip/1e9:.1f} direct-conv-equivalent TFLOP/s: a speed figure -- the Winograd layers issue 4/9 or 1/4 of these FLOPs) | MIOpen {t_mi:.2f} ms; "
          f"decoder {td:.2f} ms; whole model eager {te:.2f} ms, hipGraph {tg:.2f} ms; "
          f"with the split-bf16 decoder: decoder {td3:.2f} ms, whole model {te3:.2f} ms; "
          f"with split-bf16 3x3 encoder layers too: encoder {t_hip3:.2f} ms, whole model {te33:.2f} ms", flush=True)
