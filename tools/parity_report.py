"""Print max|HIP - reference| per golden case for both sine modes (GPU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import diinn_amd.synth as synth, diinn_amd.decoder as D
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import diinn_oracle as orc

g = np.load(os.path.join(ROOT, "tests", "golden", "diinn_golden.npz"))
dev = torch.device("cuda:0")
for k in g.files:
    if not k.startswith("meta/"):
        continue
    name = k[5:]
    b, h, w, hu, wu, gain, _ = g[k]
    b, h, w, hu, wu = map(int, (b, h, w, hu, wu))
    sd = synth.decoder_state_dict(123, float(gain))
    packed = D.pack_state_dict(sd).to(dev)
    feat = torch.from_numpy(synth.encoder_features(123, b, h, w)).to(dev)
    ref = g["out/" + name]
    truth = orc.decode_reference_form_f64(sd, synth.encoder_features(123, b, h, w), (hu, wu)).numpy()
    errs, errs64 = [], []
    for mode in (0, 1, 2):
        out = D.decode_features(feat, packed, (hu, wu), sin_mode=mode).cpu().numpy()
        errs.append(float(np.abs(out - ref).max()))
        errs64.append(float(np.abs(out - truth).max()))
    print(f"{name:26s} max|ref|={np.abs(ref).max():.3f}  vs reference(fp32): accurate {errs[0]:.1e} hw {errs[1]:.1e} "
          f"hw_reduced {errs[2]:.1e} | vs float64 truth: reference {np.abs(ref - truth).max():.1e} "
          f"hip accurate {errs64[0]:.1e} hw {errs64[1]:.1e} hw_reduced {errs64[2]:.1e}")
