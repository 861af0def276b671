import json, os, sys
import torch
from toda_amd import ops
from toda_amd.tools.bench_conv2d import timed
res = {}
for name, b, cin, cout, h, w in (("128@188", 2, 128, 128, 188, 188), ("256@94", 2, 256, 256, 94, 94), ("64->320@188", 2, 64, 320, 188, 188)):
    x = torch.relu(torch.randn(b, cin, h, w, device="cuda"))
    wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    u0 = ops.conv3x3_transform_weight(wt, 0)
    res[name] = round(timed(lambda: ops.conv3x3_run(x, u0, None, cout), 50) * 1e3, 1)
print(os.environ.get("TODA_WINO_ABLATE", "0"), json.dumps(res), flush=True)
