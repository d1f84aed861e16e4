import sys, ctypes as C
sys.path.insert(0, '/root/repo')
import torch, numpy as np
import diinn_amd._native as N, diinn_amd.decoder as D, diinn_amd.synth as synth
lib = N.load(); dev = torch.device('cuda:0')
sd = synth.decoder_state_dict(123)
packed = D.pack_state_dict(sd).to(dev)
for hw in (48, 64, 128, 256):
    feat = torch.from_numpy(synth.encoder_features(1, 1, hw, hw)).to(dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P1 = torch.zeros(hw * hw * 1024, device=dev); P2 = torch.zeros_like(P1)
    N.check(lib.diinn_precompute_P(st, C.c_void_p(feat.data_ptr()), C.c_void_p(packed.data_ptr()), C.c_void_p(P1.data_ptr()), 1, hw, hw, 0, hw), 'P')
    N.check(lib.diinn_precompute_P_ex(st, C.c_void_p(feat.data_ptr()), C.c_void_p(packed.data_ptr()), C.c_void_p(P2.data_ptr()), 1, hw, hw, 0, hw, 0), 'Pex')
    torch.cuda.synchronize()
    print(hw, 'P direct nan', int(torch.isnan(P1).sum()), 'P wino nan', int(torch.isnan(P2).sum()), 'maxdiff', float((P1 - P2).abs().max()))
    for k in (1, 2):
        N.debug_set('DIINN_F32_KERNEL', k)
        out = D.decode_features(feat, packed, (4 * hw, 4 * hw))
        torch.cuda.synchronize()
        print('   decode kernel', k, 'nan', int(torch.isnan(out).sum()), 'of', out.numel(), 'absmax', float(out[~torch.isnan(out)].abs().max()) if (~torch.isnan(out)).any() else None)
    N.debug_set('DIINN_F32_KERNEL', 0)
