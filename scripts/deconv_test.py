import torch, torch.nn.functional as F
from toda_amd.tools.bench_conv2d import timed
torch.manual_seed(0)
def run(name, f, params):
    def step():
        for p in params: p.grad = None
        y = f()
        y.backward(g[name])
    return timed(step, 30)
g = {}
# level 0: ConvTranspose2d(128, 256, 1, stride 1) on [2,128,188,188]
x0 = torch.randn(2, 128, 188, 188, device="cuda", requires_grad=True)
w0 = (torch.randn(128, 256, 1, 1, device="cuda") * 0.05).requires_grad_(True)
g["a"] = torch.randn(2, 256, 188, 188, device="cuda"); g["b"] = g["a"]
ta = run("a", lambda: F.conv_transpose2d(x0, w0, stride=1), [x0, w0])
ga = (x0.grad.clone(), w0.grad.clone()); ya = F.conv_transpose2d(x0, w0, stride=1)
tb = run("b", lambda: F.conv2d(x0, w0.transpose(0, 1)), [x0, w0])
yb = F.conv2d(x0, w0.transpose(0, 1))
print("k1: conv_transpose %.3f ms, conv1x1 %.3f ms, max diff y %.2e dx %.2e dw %.2e" % (ta, tb, float((ya-yb).abs().max()), float((ga[0]-x0.grad).abs().max()), float((ga[1]-w0.grad).abs().max())))
# level 1: ConvTranspose2d(256, 256, 2, stride 2) on [2,256,94,94]
x1 = torch.randn(2, 256, 94, 94, device="cuda", requires_grad=True)
w1 = (torch.randn(256, 256, 2, 2, device="cuda") * 0.05).requires_grad_(True)
g["c"] = torch.randn(2, 256, 188, 188, device="cuda"); g["d"] = g["c"]
tc = run("c", lambda: F.conv_transpose2d(x1, w1, stride=2), [x1, w1])
gc = (x1.grad.clone(), w1.grad.clone()); yc = F.conv_transpose2d(x1, w1, stride=2)
def ps():
    wk = w1.permute(1, 2, 3, 0).reshape(256 * 4, 256, 1, 1)      # [(co,dy,dx), ci]
    return F.pixel_shuffle(F.conv2d(x1, wk), 2)
td = run("d", ps, [x1, w1])
yd = ps()
print("k2s2: conv_transpose %.3f ms, conv1x1+pixel_shuffle %.3f ms, max diff y %.2e dx %.2e dw %.2e" % (tc, td, float((yc-yd).abs().max()), float((gc[0]-x1.grad).abs().max()), float((gc[1]-w1.grad).abs().max())))
