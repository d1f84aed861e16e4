"""RDN trunk with its 3x3 layers on Winograd F(2x2,3x3) vs F(4x4,3x3) (forced on), what the dispatch rule picks, and both
against MIOpen.  usage: enc_wino4_trunk_ab.py [SIZE | BxHxW ...]"""
import sys, torch
sys.path.insert(0, '/root/repo')
import diinn_amd.modules as M
dev = torch.device("cuda:0")
enc = M.make_rdn().to(dev).eval()
def t_ms(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
with torch.no_grad():
    from diinn_amd import _native
    lib = _native.load()
    for arg in sys.argv[1:] or ["192", "256", "384"]:
        bb, hh, ww = (int(v) for v in arg.split("x")) if "x" in arg else (1, int(arg), int(arg))
        lr = arg
        x = torch.rand(bb, 3, hh, ww, device=dev)
        pick = lib.diinn_rdn_wino4_applies(bb, hh, ww)
        enc.hip_winograd4 = False; a = enc(x); ta = t_ms(lambda: enc(x))
        _native.debug_set("DIINN_ENC_WINO4_MIN", 0)
        enc.hip_winograd4 = True; b = enc(x); tb = t_ms(lambda: enc(x))
        _native.debug_set("DIINN_ENC_WINO4_MIN", -1)
        lr = f"{arg} (rule: {'F(4,3)' if pick else 'F(2,3)'}{'' if (tb < ta) == bool(pick) else '  <-- WRONG'})"
        enc.hip_trunk_max_pixels = None; r = enc(x); enc.hip_trunk_max_pixels = M.RDN.hip_trunk_max_pixels
        print(f"{lr}: F(2,3) {ta:.3f} ms  F(4,3) {tb:.3f} ms   |F23-miopen| {(a-r).abs().max().item():.2e} |F43-miopen| {(b-r).abs().max().item():.2e} max|ref| {r.abs().max().item():.3f}", flush=True)
