"""bds_nonfinite_flags_kinds over the 472 MB of a 2 M-Gaussian parameter set: plain vs activation-aware kinds (us)."""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bilateral_driving_amd import _lib as L
lib = L.lib(); N = 2_000_000
ts = [torch.randn(N, 3).cuda(), torch.randn(N, 4).cuda(), torch.randn(N, 3).cuda(), torch.randn(N, 1).cuda(), torch.randn(N, 3).cuda(), torch.randn(N, 15, 3).cuda()]
flag = torch.zeros(1, device="cuda", dtype=torch.int32)
ptrs = (ctypes.c_void_p * 6)(*[t.data_ptr() for t in ts]); cnts = (ctypes.c_int64 * 6)(*[t.numel() for t in ts])
def run(kinds):
    k = None if kinds is None else (ctypes.c_int * 6)(*kinds)
    for _ in range(3): lib.bds_nonfinite_flags_kinds(6, ptrs, cnts, k, flag.data_ptr(), None, L.stream())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): lib.bds_nonfinite_flags_kinds(6, ptrs, cnts, k, flag.data_ptr(), None, L.stream())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3
print("plain", run(None), "kinds", run((0, 2, 1, 3, 0, 0)), "bytes", sum(t.numel() for t in ts) * 4 / 1e6, "MB")
