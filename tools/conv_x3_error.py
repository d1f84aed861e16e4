"""CPU experiment: one 3x3 convolution layer of the encoder in split-bf16 arithmetic (direct form, hi/lo bf16 operands, three
products) against float64, beside the direct fp32 sum and the fp32 Winograd F(2x2,3x3) form the trunk uses.  usage: python tools/conv_x3_error.py"""
import torch, numpy as np, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import diinn_oracle as orc
torch.manual_seed(0)
bf = orc._bf16_round
def split(t):
    hi = bf(t); return hi, bf(t - hi)
def conv_x3(x, w):
    # x [1,C,H,W], w [O,C,3,3]: unfold -> matmul with 3 bf16 products, fp32 accumulate
    u = torch.nn.functional.unfold(x, 3, padding=1)[0]          # [C*9, HW]
    wm = w.reshape(w.shape[0], -1)                              # [O, C*9]
    uh, ul = split(u); wh, wl = split(wm)
    return (wl @ uh + wh @ ul) + wh @ uh
def wino_f23(x, w):
    # F(2x2,3x3) in fp32 (float64 transforms of weights rounded once, as the kernel)
    G = torch.tensor([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], dtype=torch.float64)
    Bt = torch.tensor([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]], dtype=torch.float32)
    At = torch.tensor([[1,1,1,0],[0,1,-1,-1]], dtype=torch.float32)
    U = (G @ w.double() @ G.t()).float()                        # [O,C,4,4]
    C, H, W = x.shape[1:]
    xp = torch.nn.functional.pad(x[0], (1,1,1,1))
    out = torch.zeros(w.shape[0], H, W)
    for ty in range(0, H, 2):
        tiles = torch.stack([xp[:, ty:ty+4, tx:tx+4] for tx in range(0, W, 2)], 0)   # [T,C,4,4]
        V = Bt @ tiles @ Bt.t()
        M = torch.einsum('ocij,tcij->toij', U, V)
        Y = At @ M @ At.t()                                      # [T,O,2,2]
        for k, tx in enumerate(range(0, W, 2)):
            out[:, ty:ty+2, tx:tx+2] = Y[k]
    return out
for C in (64, 256, 512):
    O, H, W = 64, 16, 16
    x = torch.randn(1, C, H, W) * 0.5
    x = torch.relu(x) if C > 64 else x
    w = (torch.rand(O, C, 3, 3) * 2 - 1) / (C * 9) ** 0.5 * 1.7
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)[0]
    d32 = torch.nn.functional.conv2d(x, w, padding=1)[0]
    x3 = conv_x3(x, w).reshape(O, H, W)
    wf = wino_f23(x, w)
    sc = ref.abs().max().item()
    print(f"C={C}: |out| {sc:.2f}  direct fp32 {((d32-ref).abs().max()/sc):.2e}  F(2,3) fp32 {((wf-ref).abs().max()/sc):.2e}  bf16x3 direct {((x3-ref).abs().max()/sc):.2e}  rms: fp32 {((d32-ref).pow(2).mean().sqrt()/sc):.2e} F23 {((wf-ref).pow(2).mean().sqrt()/sc):.2e} x3 {((x3-ref).pow(2).mean().sqrt()/sc):.2e}")
