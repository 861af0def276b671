"""The two kernels whose workgroups wait for each other inside one launch (bn2d split kernels, Winograd stream-K forward) on a
stream restricted to a quarter of the CUs: their grids (one workgroup per CU of the whole chip / one per plane) are
then larger than what is resident, which is the situation their "partner is resident or next in line" argument has to survive.
The spins are bounded (toda_device_fault), so a broken argument fails this test instead of hanging the GPU."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _masked_stream(n_cus):
    hip = ctypes.CDLL("libamdhip64.so")
    words = (ctypes.c_uint32 * 8)(*([0] * 8))           # 256 CUs; enable every 4th CU, spread over all XCDs
    for cu in range(256):
        if cu % (256 // n_cus) == 0:
            words[cu // 32] |= 1 << (cu % 32)
    stream = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(stream), ctypes.c_uint32(8), words)
    if rc != 0 or not stream.value:
        pytest.skip(f"hipExtStreamCreateWithCUMask unavailable (rc {rc})")
    return hip, stream


def test_inter_workgroup_waits_survive_a_quarter_of_the_cus():
    from toda_amd import lib as L
    from toda_amd import ops

    lib = L.load()
    assert lib.toda_device_fault() == 0
    torch.manual_seed(0)
    x = torch.randn(2, 128, 188, 188, device="cuda")
    w = torch.randn(128, 128, 3, 3, device="cuda") * 0.05
    bn = torch.nn.BatchNorm2d(128).cuda().train()
    gy = torch.randn(2, 128, 188, 188, device="cuda")

    def run():
        xx = x.clone().requires_grad_(True)
        bn.running_mean.zero_(); bn.running_var.fill_(1.0)
        y = ops.conv3x3(xx, w)                       # wino_fwd_ws_kernel forward, and again (rotated filters) in backward
        z = ops.bn2d(y, bn, True)                    # per-plane backward kernel at this shape (partner exchange)
        z.backward(gy)
        return y.detach().clone(), z.detach().clone(), xx.grad.clone(), bn.weight.grad.clone()

    bn.weight.grad = None
    ref = run()
    torch.cuda.synchronize()
    assert lib.toda_device_fault() == 0
    hip, raw = _masked_stream(64)
    try:
        ext = torch.cuda.ExternalStream(raw.value)
        ext.wait_stream(torch.cuda.current_stream())
        bn.weight.grad = None
        with torch.cuda.stream(ext):
            got = run()
        ext.synchronize()
        rc = lib.toda_device_fault()
        assert rc == 0, lib.toda_last_error()
        for a, b in zip(ref, got):
            assert torch.equal(a, b)                 # same grids, same partitions, same summation order: the same bits
    finally:
        torch.cuda.synchronize()
        hip.hipStreamDestroy(raw)
