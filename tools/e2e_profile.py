"""Whole DIINN forward (RDN encoder + implicit decoder) at one size, eager, for rocprofv3 --kernel-trace --stats:
    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 tools/e2e_profile.py [LR] [SCALE] [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import diinn_amd.modules as M

lr = int(sys.argv[1]) if len(sys.argv) > 1 else 256
s = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda:0")
net = M.DIINN(mode=3, init_q=False).to(dev).eval()
x = torch.rand(1, 3, lr, lr, device=dev)
with torch.no_grad():
    for _ in range(n):
        y = net(x, (lr * s, lr * s), 30000)
torch.cuda.synchronize()
print(tuple(y.shape))
