import sys, os, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import diinn_amd.decoder as D, diinn_amd.synth as synth, diinn_amd.training as T
dev = torch.device("cuda:0")
dec = D.ImplicitDecoder(mode=3, init_q=False)
dec.load_state_dict({k: torch.from_numpy(v) for k, v in synth.decoder_state_dict(123).items()})
dec = dec.to(dev).train()
feat = torch.from_numpy(synth.encoder_features(123, 16, 48, 48)).to(dev).requires_grad_(True)
r = torch.randn(16, 3, 192, 192, device=dev)
def step():
    dec.zero_grad(set_to_none=True); feat.grad = None
    (dec(feat, [192, 192]) * r).sum().backward()
for rs in (1024, 512, 256, 128, 1024, 256):
    T.ROWDOT_SPLITS = rs
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8): step()
    torch.cuda.synchronize()
    print("ROWDOT_SPLITS", rs, "step %.3f ms" % ((time.perf_counter() - t0) / 8 * 1e3))
