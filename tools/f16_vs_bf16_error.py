"""CPU experiment: fp16 instead of bf16 operands in the optional reduced-precision paths (same MFMA rate on gfx950),
emulated in the oracle against the fp32 reference.  usage: python tools/f16_vs_bf16_error.py"""
import sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import diinn_oracle as orc, diinn_amd.synth as synth
def rnd16(t, kind):
    return t.to(torch.float16).to(torch.float32) if kind == "f16" else orc._bf16_round(t)
def decode16(sd, feat, size, kind, full):
    sw = orc.split_weights(sd); feat = orc._as_t(feat); b, c, h, w = feat.shape; hu, wu = size
    small = orc.uses_small_output_kernel(hu, wu)
    idx_h, rel_h = orc.axis_tables(h, hu, small); idx_w, rel_w = orc.axis_tables(w, wu, small)
    if full:
        import torch.nn.functional as F
        p = F.conv2d(rnd16(feat, kind), rnd16(sw["Wx"].view(1024, 64, 3, 3), kind), None, padding=1)
        p = (p + sw["bK"].view(1, -1, 1, 1)).permute(0, 2, 3, 1).contiguous()
    else:
        p = orc.precompute_P(sd, feat)
    pp = p[:, torch.from_numpy(idx_h.astype(np.int64))][:, :, torch.from_numpy(idx_w.astype(np.int64))].view(b, hu, wu, 4, 256)
    syn = torch.empty((hu, wu, 3)); syn[..., 0] = torch.from_numpy(rel_h)[:, None]; syn[..., 1] = torch.from_numpy(rel_w)[None, :]; syn[..., 2] = float(orc.scale_ratio(h, w, hu, wu))
    q = torch.relu(pp[:, :, :, 0]) * torch.sin(syn @ sw["Q0"].t() + sw["bQ"][0])
    inv = torch.tensor(0.15915494309189533577)
    for i in range(1, 4):
        qi = rnd16(q, kind)
        k = torch.relu(qi @ rnd16(sw["Wq"][i-1], kind).t() + pp[:, :, :, i])
        rev = qi @ rnd16(sw["Qw"][i-1] * inv, kind).t() + sw["bQ"][i] * inv
        q = k * torch.sin(rev.double() * (2 * np.pi)).float()
    return (q @ sw["L"].t() + sw["bL"]).permute(0, 3, 1, 2).contiguous()
print("max|err| / max|ref|   (restated bounds: bf16 2e-3, bf16_full 3e-3)")
for gain in (1.0, 3.0):
    for (b,h,w,hu,wu,seed) in [(1,48,48,96,96,123),(1,40,56,132,185,7),(1,64,64,256,256,123)]:
        sd = synth.decoder_state_dict(seed, gain); feat = synth.encoder_features(seed, b, h, w)
        ref = orc.decode_reference_form(sd, feat, (hu,wu), 30000).numpy(); sc = np.abs(ref).max()
        r = {}
        for kind in ("bf16", "f16"):
            for full in (False, True):
                r[(kind, full)] = np.abs(decode16(sd, feat, (hu,wu), kind, full).numpy() - ref).max() / sc
        print(f"gain {gain} {h}x{w}->{hu}x{wu} |ref| {sc:.3f}: bf16 {r[('bf16',False)]:.1e}  bf16_full {r[('bf16',True)]:.1e}  |  f16 {r[('f16',False)]:.1e}  f16_full {r[('f16',True)]:.1e}   (fp32 tolerance as relative: {1e-4*max(1,sc)/sc:.1e})")
