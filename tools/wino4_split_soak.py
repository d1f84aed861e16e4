"""Soak of the split F(4x4,3x3) layer's slab hand-off: N launches per shape on alternating inputs through ONE workspace, a second
stream decoding meanwhile (uneven load), every output compared bit for bit with the first on its input; counter words at the end.
usage: wino4_split_soak.py [N=1500]"""
import ctypes as C, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import diinn_amd._native as N, diinn_amd.decoder as D, diinn_amd.modules as M, diinn_amd.synth as synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
dev = torch.device("cuda:0"); lib = N.load()
gen = torch.Generator(device=dev).manual_seed(5)
ptr = lambda t: C.c_void_p(t.data_ptr())
wsf = lib.diinn_conv_wino4_workspace_floats(); ws = torch.zeros(wsf, device=dev)
N.debug_set("DIINN_ENC_WINO4_SPLIT", 2)
packed_dec = D.pack_state_dict(synth.decoder_state_dict(3)).to(dev)
feat = torch.from_numpy(synth.encoder_features(3, 1, 64, 64)).to(dev)
side = torch.cuda.Stream(device=dev); main = torch.cuda.current_stream(dev)
for (b, cin, h, w) in [(1, 512, 192, 192), (1, 256, 100, 100), (1, 384, 320, 320), (2, 128, 200, 180), (1, 64, 144, 144)]:
    xs = [torch.randn((b, cin, h, w), device=dev, generator=gen) for _ in range(2)]
    wt = torch.randn((64, cin, 3, 3), device=dev, generator=gen) / (cin * 9) ** 0.5
    bias = torch.randn(64, device=dev, generator=gen); pk = M.pack_conv_wino4(wt).to(dev)
    outs = [torch.empty((b, 64, h, w), device=dev) for _ in range(2)]
    first = [None, None]; bad = 0; t0 = time.time()
    info = (C.c_int * 4)(); lib.diinn_conv_wino4_plan(cin, b, h, w, 1, info)
    for i in range(n):
        k = i & 1
        if i % 7 == 0:
            with torch.cuda.stream(side):
                D.decode_features(feat, packed_dec, (96 + 16 * (i % 11), 160))
        assert lib.diinn_conv_wino4_ws(C.c_void_p(main.cuda_stream), ptr(xs[k]), cin * h * w, cin, ptr(pk), ptr(bias), None, 0, ptr(outs[k]), 64 * h * w,
                                       1, b, h, w, ptr(ws), wsf) == 0
        if first[k] is None:
            first[k] = outs[k].clone()
        elif not torch.equal(outs[k], first[k]):
            bad += 1
    torch.cuda.synchronize()
    nz = int(ws[:1024].view(torch.int32).abs().sum())
    print(f"{b}x{cin}x{h}x{w}: plan {list(info)}, {n} launches on alternating inputs, {bad} differ from the first on their input; counter words non-zero: {nz}; {time.time() - t0:.1f} s", flush=True)
