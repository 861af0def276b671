"""Times toda_conv3x3_wgrad alone (per launch group of three kernels) at the C3 neck shapes."""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from toda_amd import ops
torch.manual_seed(0)
shapes = [(2, 256, 256, 94, 94), (2, 128, 128, 188, 188), (2, 64, 512, 188, 188), (2, 128, 256, 188, 188)]
n = int(os.environ.get("N_IT", "30"))
for (B, ci, co, H, W) in shapes:
    x = torch.randn(B, ci, H, W, device="cuda")
    gy = torch.randn(B, co, H, W, device="cuda")
    for it in range(3):
        dw = ops.conv3x3_wgrad(x, gy, (co, ci, 3, 3))
    torch.cuda.synchronize()
    e0, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
    e0.record()
    for _ in range(n):
        dw = ops.conv3x3_wgrad(x, gy, (co, ci, 3, 3))
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / n
    fl = 2.0 * B * H * W * ci * co * 9
    err = ""
    if os.environ.get("CHECK"):
        ref = torch.nn.grad.conv2d_weight(x.double(), (co, ci, 3, 3), gy.double(), padding=1)
        err = f" rel err {float((dw.double() - ref).abs().max() / ref.abs().max()):.2e}"
    print((B, ci, co, H, W), f"wgrad {t*1e3:.1f} us ({fl/t/1e9:.1f} TF/s direct-eq, {fl/4/t/1e9:.1f} TF/s issued)" + err, flush=True)
