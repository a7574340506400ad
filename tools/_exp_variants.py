import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tools"))
import bench_kernels as bk
from dkt_stereo_amd import conv
conv.set_backend("f16x3")
name = os.path.basename(os.environ["DKT_LIB_PATH"])
with torch.no_grad():
    for cout, cin, H, W in ((256, [128,128,128], 184, 312), (128, [128,128,128], 184, 312), (64, [64], 736, 1248), (256, [128,128,128], 92, 156)):
        layer = torch.nn.Conv2d(sum(cin), cout, 3, padding=1).to("cuda:0")
        xs = [torch.randn(1, c, H, W, device="cuda:0") for c in cin]
        us = bk.timeit(lambda: conv.conv2d(xs, layer), n=20, warm=3)
        print("%%-18s cout=%%3d cin=%%3d %%dx%%d  %%8.1f us" %% (name, cout, sum(cin), H, W, us), flush=True)
''' % (ROOT, ROOT)
for lib in sorted(glob.glob(os.path.join(ROOT, "dkt_stereo_amd/lib/variants/lib_*.so"))):
    env = dict(os.environ, DKT_LIB_PATH=lib)
    subprocess.run([sys.executable, "-c", code], env=env)
