import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tools')
import bench_kernels as bk
from dkt_stereo_amd import conv
conv.set_backend("f16x3")
with torch.no_grad():
    for cout in (256, 128):
        layer = torch.nn.Conv2d(384, cout, 3, padding=1).to("cuda:0")
        xs = [torch.randn(1, 128, 184, 312, device="cuda:0") for _ in range(3)]
        for nf in ("4", "2"):
            os.environ["DKT_CONV_NF"] = nf
            bk.report("cout=%d NF=%s" % (cout, nf), bk.timeit(lambda: conv.conv2d(xs, layer), n=20, warm=3))
