"""diinn_conv_wino4 launched repeatedly on the same input: every result must be bit-identical (an LDS-DMA read placed in the
wrong barrier phase shows up as rare differing tiles, not as a failed tolerance check) and equal a float64 convolution to the
kernel's tolerance.  usage: conv_wino4_determinism.py [H W CIN REPEATS]"""
import ctypes as C, sys, torch, torch.nn.functional as F
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import diinn_amd._native as N, diinn_amd.modules as M
h, w, cin, reps = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (256, 256, 512, 200)
dev = torch.device("cuda:0"); lib = N.load()
gen = torch.Generator(device=dev).manual_seed(1)
x = torch.rand((1, cin, h, w), device=dev, generator=gen)
wt = (torch.rand((64, cin, 3, 3), device=dev, generator=gen) - 0.5)
bias = torch.zeros(64, device=dev)
pk = M.pack_conv_wino4(wt).to(dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ptr = lambda t: C.c_void_p(t.data_ptr())
outs = [torch.empty((1, 64, h, w), device=dev) for _ in range(4)]
def run(o):
    assert lib.diinn_conv_wino4(st, ptr(x), cin * h * w, cin, ptr(pk), ptr(bias), None, 0, ptr(o), 64 * h * w, 1, 1, h, w) == 0
run(outs[0]); torch.cuda.synchronize()
ref = torch.relu(F.conv2d(x.double(), wt.double(), None, padding=1))
print("vs float64: max err %.3e of max|ref| %.1f" % (float((outs[0].double() - ref).abs().max()), float(ref.abs().max())))
bad = 0
for i in range(reps):
    o = outs[1 + i % 3]
    run(o)                                                       # back to back, no synchronisation in between
    if i % 3 == 2:
        torch.cuda.synchronize()
        for oo in outs[1:]:
            d = (oo != outs[0])
            if bool(d.any()):
                bad += 1
                idx = d.nonzero()
                print("launch group %d: %d differing outputs, first at %s, rows %d..%d" % (i, int(d.sum()), idx[0].tolist(), int(idx[:, 2].min()), int(idx[:, 2].max())))
print("differing launches: %d of %d" % (bad, reps))
