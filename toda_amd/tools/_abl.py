import torch, sys, os
from toda_amd import ops
from toda_amd.tools.bench_conv2d import timed
b,cin,cout,h,w = 2,128,128,188,188
x=torch.relu(torch.randn(b,cin,h,w,device="cuda")); wt=torch.randn(cout,cin,3,3,device="cuda")*0.03
u0=ops.conv3x3_transform_weight(wt,0)
print(os.environ.get("TODA_WINO_ABLATE"), os.environ.get("TODA_WINO_VARIANT"), round(timed(lambda: ops.conv3x3_run(x,u0,None,cout), 20)*1e3,1), "us")
