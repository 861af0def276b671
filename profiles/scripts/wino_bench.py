import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from toda_amd import ops
torch.manual_seed(0)
shapes = [(2, 256, 256, 94, 94), (2, 128, 128, 188, 188)]
if os.environ.get("WINO_SHAPE"):          # one shape per process: the persistent kernels' grid is the CU count for every shape,
    shapes = [shapes[int(os.environ["WINO_SHAPE"])]]      # so a counter pass can only tell shapes apart when a run holds one
n = int(os.environ.get("N_IT", "20"))
for (B, ci, co, H, W) in shapes:
    x = torch.randn(B, ci, H, W, device="cuda", requires_grad=True)
    w = (torch.randn(co, ci, 3, 3, device="cuda") * 0.05).requires_grad_(True)
    gy = torch.randn(B, co, H, W, device="cuda")
    for it in range(3):
        y = ops.conv3x3(x, w); y.backward(gy)
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    ys = [ops.conv3x3(x, w) for _ in range(n)]
    e1.record()
    for y in ys:
        y.backward(gy, retain_graph=False)
    e2.record(); torch.cuda.synchronize()
    fl = 2.0 * B * H * W * ci * co * 9
    f, b = e0.elapsed_time(e1) / n, e1.elapsed_time(e2) / n
    print((B, ci, co, H, W), f"fwd {f*1e3:.1f} us ({fl/f/1e9:.1f} TF/s direct-eq)  bwd(dgrad+wgrad) {b*1e3:.1f} us")
