import json, os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import diinn_amd.modules as M, diinn_amd.synth as synth
gold = np.load('/root/repo/tests/golden/rdn_big_golden.npz')
dev = torch.device("cuda:0")
enc = M.make_rdn()
shapes = json.loads(str(gold["rdn/shapes_json"]))
enc.load_state_dict({k: torch.from_numpy(v) for k, v in synth.state_dict_for(shapes, 123, "enc.").items()})
enc = enc.to(dev).eval()
for w4 in (True, False):
    enc.hip_winograd4 = w4
    for (b, h, w) in [(1, 96, 100), (1, 240, 256), (2, 50, 90)]:
        key = f"{b}x{h}x{w}"
        x = torch.from_numpy(synth.uniform(7, f"img:{b}x{h}x{w}", (b, 3, h, w), 0.5) + np.float32(0.5)).to(dev)
        with torch.no_grad():
            y = enc(x).cpu().numpy()
        idx = np.random.default_rng(1000 * h + w).choice(y.size, size=min(16384, y.size), replace=False)
        scale = max(1.0, float(gold[f"rdn/{key}/absmax"]))
        err = float(np.abs(y.reshape(-1)[idx] - gold[f"rdn/{key}/values"]).max())
        print(f"hip_winograd4={w4} {key}: err {err:.3e} = {err / (2e-5 * scale):.3f} of the bound (absmax {scale:.2f})")
