import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dkt_stereo_amd.submodule import _group_l2norm, build_gwc_volume, build_norm_correlation_volume
DEV="cuda:0"
def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n
with torch.no_grad():
    a, b = (torch.randn(1, 96, 184, 312, device=DEV) for _ in range(2))
    print("l2norm G=1 96ch: %.1f us" % timeit(lambda: _group_l2norm(a, 1)))
    na, nb = _group_l2norm(a, 1), _group_l2norm(b, 1)
    print("gwc G=1 cpg=96 D=48: %.1f us" % timeit(lambda: build_gwc_volume(na, nb, 48, 1)))
    print("gwc G=8 cpg=12 D=48: %.1f us" % timeit(lambda: build_gwc_volume(na, nb, 48, 8)))
    print("gwc G=6 cpg=16 D=48: %.1f us" % timeit(lambda: build_gwc_volume(na, nb, 48, 6)))
    print("total: %.1f us" % timeit(lambda: build_norm_correlation_volume(a, b, 48)))
