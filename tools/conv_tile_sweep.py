"""Tile-shape sweep (DKT_CONV_CFG, read once per process) for the coarse-GRU and small-layer convolution shapes."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tools"))
from iteration_stages import gtime
from dkt_stereo_amd import conv
CASES = (("gru16.zr", 256, [128, 128, 128], 92, 156), ("gru16.q", 128, [128, 128, 128], 92, 156),
         ("gru32.zr", 256, [128, 128], 46, 78), ("gru32.q", 128, [128, 128], 46, 78),
         ("enc.conv", 126, [64, 64], 184, 312), ("convc2", 64, [64], 184, 312), ("fh.conv1", 256, [128], 184, 312),
         ("gru08.q", 128, [128, 128, 128], 184, 312))
with torch.no_grad():
    row = []
    for nm, cout, cin, H, W in CASES:
        layer = torch.nn.Conv2d(sum(cin), cout, 3, padding=1).to("cuda:0")
        xs = [torch.randn(1, c, H, W, device="cuda:0") for c in cin]
        try:
            us = gtime(lambda: conv.conv2d(xs, layer, relu=True))
        except Exception as e:
            us = float("nan")
        row.append("%%s %%6.1f" %% (nm, us))
    print("cfg=%%-4s " %% os.environ.get("DKT_CONV_CFG", "auto") + " | ".join(row), flush=True)
''' % (ROOT, ROOT)
for cfg in (None, "1", "2", "3", "5", "6", "7", "10"):
    env = dict(os.environ)
    if cfg:
        env["DKT_CONV_CFG"] = cfg
    subprocess.run([sys.executable, "-c", code], env=env)
