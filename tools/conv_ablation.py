"""Times the update block's convolution shapes with every library under dkt_stereo_amd/lib/variants
(tools/build_abl_variants.sh: ablation builds) and the product library."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tools"))
import bench_kernels as bk
from dkt_stereo_amd import conv
conv.set_backend("f16x3")
name = os.path.basename(os.environ.get("DKT_LIB_PATH", "product"))
import os as _os
CASES = (("s2 64>96@736", 96, [64], 3, 736, 1248, 2), ("s2 64>96@736 B2", 96, [64], 3, 736, 1248, 2, 2), ("s2 96>128@368", 128, [96], 3, 368, 624, 2), ("s2 96>128@368 B2", 128, [96], 3, 368, 624, 2, 2),
         ("1x1s2 64>96@736", 96, [64], 1, 736, 1248, 2), ("1x1s2 96>128@368", 128, [96], 1, 368, 624, 2), ("s2 128>128@184", 128, [128], 3, 184, 312, 2)) if _os.environ.get("S2") else (("enc64@736", 64, [64], 3, 736, 1248), ("enc96@368", 96, [96], 3, 368, 624), ("enc128@184", 128, [128], 3, 184, 312)) if _os.environ.get("ENC") else (("convc1", 64, [36], 1, 184, 312), ("convc2", 64, [64], 3, 184, 312), ("enc.conv", 126, [64, 64], 3, 184, 312),
         ("gru08.zr", 256, [128, 128, 128], 3, 184, 312), ("gru08.q", 128, [128, 128, 128], 3, 184, 312),
         ("fh.conv1", 256, [128], 3, 184, 312), ("fh.conv2", 2, [256], 3, 184, 312),
         ("gru16.zr", 256, [128, 128, 128], 3, 92, 156), ("gru32.zr", 256, [128, 128], 3, 46, 78))
with torch.no_grad():
    row = []
    for case in CASES:
        nm, cout, cin, k, H, W = case[:6]
        stride = case[6] if len(case) > 6 else 1
        nb = case[7] if len(case) > 7 else 1
        layer = torch.nn.Conv2d(sum(cin), cout, k, padding=k // 2, stride=stride).to("cuda:0")
        xs = [torch.randn(nb, c, H, W, device="cuda:0") for c in cin]
        fn = lambda: conv.conv2d(xs, layer, relu=True)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(10): fn()
        us = bk.timeit(g.replay, n=10, warm=2) / 10.0
        row.append("%%s %%6.1f" %% (nm, us))
    print("%%-14s " %% name + " | ".join(row), flush=True)
''' % (ROOT, ROOT)
libs = [None] + sorted(glob.glob(os.path.join(ROOT, "dkt_stereo_amd/lib/variants/lib_*.so")))
for lib in libs:
    env = dict(os.environ)
    if lib:
        env["DKT_LIB_PATH"] = lib
        env["DKT_ALLOW_ABLATION"] = "1"          # timing-only builds (the loader refuses them otherwise)
    subprocess.run([sys.executable, "-c", code], env=env)
