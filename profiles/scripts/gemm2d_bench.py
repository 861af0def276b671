import sys, os
R = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, R)
import torch
from toda_amd import ops
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
F = torch.nn.functional
for name, (kind, b, cin, cout, h, w, s) in [("conv3x3 s2 128->256 @188", ("c", 2, 128, 256, 188, 188, 0)), ("deconv 1x1 128->256 @188", ("d", 2, 128, 256, 188, 188, 1)),
                 ("deconv 2x2s2 256->256 @94", ("d", 2, 256, 256, 94, 94, 2))]:
    x = torch.randn(b, cin, h, w, device="cuda")
    if kind == "c":
        wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
        gy = torch.randn(b, cout, h // 2, w // 2, device="cuda")
        ours = lambda xx, ww: ops.conv3x3s2(xx, ww)
        ref = lambda xx, ww: F.conv2d(F.pad(xx, (1, 1, 1, 1)), ww, stride=2)
        flops = 2 * 9 * cin * cout * b * (h // 2) * (w // 2)
    else:
        wt = torch.randn(cin, cout, s, s, device="cuda") * 0.05
        gy = torch.randn(b, cout, h * s, w * s, device="cuda")
        ours = lambda xx, ww: ops.deconv(xx, ww, s)
        ref = lambda xx, ww: F.conv_transpose2d(xx, ww, stride=s)
        flops = 2 * cin * cout * s * s * b * h * w
    res = {}
    for tag, f in (("ours", ours), ("torch", ref)):
        fwd = timeit(lambda: f(x, wt))
        xg = x.clone().requires_grad_(True)
        y = f(xg, wt)
        dg = timeit(lambda: torch.autograd.grad(y, xg, gy, retain_graph=True))
        wg_ = wt.clone().requires_grad_(True)
        y2 = f(x, wg_)
        wg = timeit(lambda: torch.autograd.grad(y2, wg_, gy, retain_graph=True))
        res[tag] = (fwd, dg, wg)
    o, r = res["ours"], res["torch"]
    print(f"{name}: fwd {o[0]:.0f} / {r[0]:.0f} us, dgrad {o[1]:.0f} / {r[1]:.0f}, wgrad {o[2]:.0f} / {r[2]:.0f}  (ours / torch; {flops/1e9:.1f} GFLOP each = {flops/157.3e6:.0f} us at peak)")
