"""Full decode (precompute_P + decode kernel) time of every BASELINE.json config geometry on ONE GPU,
f32 (reference precision) and the optional bf16 paths (bf16: layers 1-3; bf16_full: the hoisted conv too).  Multi-GPU configs are run whole on one device."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import diinn_amd.decoder as D, diinn_amd.synth as synth

CONFIGS = [("c1  48x48 x2", 48, 48, 96, 96), ("c2  256x256 x4", 256, 256, 1024, 1024),
           ("c3  512x512 x4", 512, 512, 2048, 2048), ("tgt 1024x1024 x4", 1024, 1024, 4096, 4096),
           ("c4  1024x1024 x8", 1024, 1024, 8192, 8192), ("c5  720x1280 x3.3", 720, 1280, 2376, 4224)]
dev = torch.device("cuda:0")
packed = D.pack_state_dict(synth.decoder_state_dict(123)).to(dev)
print(f"{'config':20s} {'HR px':>10s} {'f32 ms':>9s} {'f32 Mpix/s':>11s} {'f32 TFLOP/s':>12s} {'bf16 ms':>9s} {'bf16 Mpix/s':>12s} "
      f"{'bf16_full ms':>13s} {'bf16_full Mpix/s':>17s}")
for name, h, w, hu, wu in CONFIGS:
    feat = torch.randn(1, 64, h, w, device=dev)
    ws = torch.empty(h * w * 1024, device=dev)
    out = torch.empty(1, 3, hu, wu, device=dev)
    res = {}
    for comp in ("f32", "bf16", "bf16_full"):
        n = 3 if hu * wu > 3e7 else 10
        for _ in range(2):
            D.decode_features(feat, packed, (hu, wu), out=out, workspace=ws, compute=comp)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            D.decode_features(feat, packed, (hu, wu), out=out, workspace=ws, compute=comp)
        e1.record(); torch.cuda.synchronize()
        res[comp] = e0.elapsed_time(e1) / n
    px = hu * wu
    flop = px * 789504.0 + h * w * 1179648.0
    print(f"{name:20s} {px:10d} {res['f32']:9.3f} {px/res['f32']/1e3:11.1f} {flop/res['f32']/1e9:12.1f} "
          f"{res['bf16']:9.3f} {px/res['bf16']/1e3:12.1f} {res['bf16_full']:13.3f} {px/res['bf16_full']/1e3:17.1f}", flush=True)
    del feat, ws, out
    torch.cuda.empty_cache()
