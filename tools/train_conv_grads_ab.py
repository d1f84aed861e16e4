"""Decoder training step (B = 16, 48x48 x4) with the hoisted 3x3 convolution's gradients on MIOpen (torch.nn.grad) or on the
library's own kernels (training.NATIVE_CONV_WGRAD / NATIVE_CONV_DGRAD)."""
import sys, torch
sys.path.insert(0, '/root/repo')
import diinn_amd.training as T, diinn_amd.modules as M
dev = torch.device("cuda:0")
torch.manual_seed(0)
dec = M.DIINN(mode=3, init_q=False).decoder.to(dev)
feat = torch.randn(16, 64, 48, 48, device=dev, requires_grad=True)
def step():
    out = dec(feat, (192, 192), None)
    out.backward(torch.ones_like(out))
def t_ms(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for rep in range(2):
    for (nw, nd) in ((False, False), (False, True), (True, True)):
        T.NATIVE_CONV_WGRAD, T.NATIVE_CONV_DGRAD = nw, nd
        print("weight gradient: %-6s  input gradient: %-6s  %.3f ms" % ("own" if nw else "MIOpen", "own" if nd else "MIOpen", t_ms(step)))
