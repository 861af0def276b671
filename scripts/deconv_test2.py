import sys, torch, torch.nn.functional as F
from toda_amd.tools.bench_conv2d import timed
torch.manual_seed(0)
mode = sys.argv[1]
x1 = torch.randn(2, 256, 94, 94, device="cuda", requires_grad=True)
w1 = (torch.randn(256, 256, 2, 2, device="cuda") * 0.05).requires_grad_(True)
g = torch.randn(2, 256, 188, 188, device="cuda")
def f_ct(): return F.conv_transpose2d(x1, w1, stride=2)
def f_c1():
    wk = w1.permute(1, 2, 3, 0).reshape(1024, 256, 1, 1)
    return F.pixel_shuffle(F.conv2d(x1, wk), 2)
def f_mm():
    wk = w1.permute(1, 2, 3, 0).reshape(1024, 256)
    return F.pixel_shuffle(torch.matmul(wk, x1.flatten(2)).view(2, 1024, 94, 94), 2)
f = {"ct": f_ct, "c1": f_c1, "mm": f_mm}[mode]
def step():
    x1.grad = None; w1.grad = None
    f().backward(g)
import time
t0 = time.time(); step(); torch.cuda.synchronize(); first = time.time() - t0
print(mode, "first call %.3f s, steady %.3f ms" % (first, timed(step, 30)))
