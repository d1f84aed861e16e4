"""Soak: 150 Adam steps of the whole network (encoder autograd on PyTorch-ROCm, decoder on the HIP forward/backward) on\nsmooth synthetic targets: the loss must stay finite and fall by more than half; then one validation step."""
import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
import diinn_amd.modules as M
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = M.SRLitModule(arch="diinn", mode=3, init_q=False).to(dev).train()
opt = torch.optim.Adam(net.parameters(), lr=1e-4)
# smooth, learnable targets: bicubic blow-ups of 6x6 noise
hr = {s: torch.nn.functional.interpolate(torch.rand(4, 3, 6, 6, device=dev), size=(24 * s, 24 * s), mode="bicubic").clamp(0, 1)
      for s in (2, 3, 4)}
lr = {s: torch.nn.functional.interpolate(hr[s], size=(24, 24), mode="bicubic", antialias=True).clamp(0, 1) for s in hr}
batch = {s: (lr[s], hr[s], None) for s in hr}
losses = []
for i in range(150):
    opt.zero_grad(set_to_none=True)
    loss = net.training_step(batch, i)["loss"]
    loss.backward()
    opt.step()
    losses.append(float(loss.detach()))
    assert np.isfinite(losses[-1]), (i, losses[-5:])
print("loss first/last:", losses[0], losses[-1], "min", min(losses), flush=True)
assert losses[-1] < 0.5 * losses[0], "training does not converge"
net.eval()
with torch.no_grad():
    res = net.validation_step(batch, 0)
print({k: float(v) for k, v in res.items()})
