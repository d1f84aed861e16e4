"""The hoisted 3x3 convolution's weight / input gradients on the library's kernels (training._conv_grads_native) against torch.nn.grad."""
import sys, torch
sys.path.insert(0, '/root/repo')
import diinn_amd.training as T
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (b, h, w) in [(1, 8, 8), (2, 8, 8), (1, 6, 5), (2, 12, 16), (16, 48, 48), (3, 7, 9)]:
    feat = torch.randn(b, 64, h, w, device=dev)
    wx = torch.randn(1024, 64, 3, 3, device=dev) * 0.05
    dp = torch.randn(b, 1024, h, w, device=dev)
    dw_ref = torch.nn.grad.conv2d_weight(feat, wx.shape, dp, padding=1).reshape(1024, 576)
    df_ref = torch.nn.grad.conv2d_input(feat.shape, wx, dp, padding=1)
    dw, df = T._conv_grads_native(feat, wx, dp, True)
    torch.cuda.synchronize()
    print((b, h, w), "dW rel err %.2e" % float((dw - dw_ref).abs().max() / dw_ref.abs().max()), "d_feat rel err %.2e" % float((df - df_ref).abs().max() / df_ref.abs().max()))
