"""CPU experiment: a split-bf16 ("bf16x3") evaluation of the per-pixel layers -- every operand as hi + lo bf16 parts, products
hi*hi + hi*lo + lo*hi on the bf16 MFMA with fp32 accumulation -- against the fp32 reference and float64.  usage: python tools/bf16x3_error.py"""
import sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import diinn_oracle as orc, diinn_amd.synth as synth
def bf(t): return orc._bf16_round(t)
def split(t):
    hi = bf(t); lo = bf(t - hi); return hi, lo
def mm3(q, w):   # q [.., K], w [M, K] -> q @ w.T with 3 bf16 products, fp32 accumulate
    qh, ql = split(q); wh, wl = split(w)
    return qh @ wh.t() + (qh @ wl.t() + ql @ wh.t())
def decode_x3(sd, feat, size):
    sw = orc.split_weights(sd)
    feat = orc._as_t(feat); b, c, h, w = feat.shape; hu, wu = size
    small = orc.uses_small_output_kernel(hu, wu)
    idx_h, rel_h = orc.axis_tables(h, hu, small); idx_w, rel_w = orc.axis_tables(w, wu, small)
    p = orc.precompute_P(sd, feat)
    pp = p[:, torch.from_numpy(idx_h.astype(np.int64))][:, :, torch.from_numpy(idx_w.astype(np.int64))].view(b, hu, wu, 4, 256)
    syn = torch.empty((hu, wu, 3)); syn[..., 0] = torch.from_numpy(rel_h)[:, None]; syn[..., 1] = torch.from_numpy(rel_w)[None, :]; syn[..., 2] = float(orc.scale_ratio(h, w, hu, wu))
    q = torch.relu(pp[:, :, :, 0]) * torch.sin(syn @ sw["Q0"].t() + sw["bQ"][0])
    for i in range(1, 4):
        k = torch.relu(mm3(q, sw["Wq"][i-1]) + pp[:, :, :, i])
        q = k * torch.sin(mm3(q, sw["Qw"][i-1]) + sw["bQ"][i])
    out = q @ sw["L"].t() + sw["bL"]
    return out.permute(0, 3, 1, 2).contiguous()
for gain in (1.0, 3.0):
    for (b,h,w,hu,wu,seed) in [(1,48,48,96,96,123),(1,40,56,132,185,7),(1,64,64,256,256,123)]:
        sd = synth.decoder_state_dict(seed, gain); feat = synth.encoder_features(seed, b, h, w)
        ref = orc.decode_reference_form(sd, feat, (hu,wu), 30000).numpy()
        r64 = orc.decode_reference_form_f64(sd, feat, (hu,wu)).numpy()
        x3 = decode_x3(sd, feat, (hu,wu)).numpy()
        hf = orc.decode_hoisted_form(sd, feat, (hu,wu)).numpy()
        print(f"gain {gain} {h}x{w}->{hu}x{wu}: max|ref|={np.abs(ref).max():.3f}  x3 vs ref {np.abs(x3-ref).max():.2e}  x3 vs f64 {np.abs(x3-r64).max():.2e}  fp32-hoisted vs f64 {np.abs(hf-r64).max():.2e}  ref32 vs f64 {np.abs(ref-r64).max():.2e}  tol {1e-4*max(1,np.abs(ref).max()):.1e}")
