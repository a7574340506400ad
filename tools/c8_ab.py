"""Interleaved A/B timing of library builds in ONE process (same tensors, alternating rounds): python tools/c8_ab.py libA.so libB.so ...
('-' = the product library).  384 -> 256 at 184x312 with cfg from C8_CFG (default 1)."""
import os, sys, torch
os.environ.setdefault("DKT_ALLOW_ABLATION", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dkt_stereo_amd import _ffi, conv_c8 as c8
from c8_check import gtime
libs = {}
for path in sys.argv[1:]:
    _ffi._lib = None
    _ffi.LIB_PATH = os.path.join(ROOT, "dkt_stereo_amd/lib/libdktstereo.so") if path == "-" else os.path.abspath(path)
    libs[path] = _ffi.lib()
torch.manual_seed(0)
cfg = int(os.environ.get("C8_CFG", "1"))
cout = int(os.environ.get("C8_COUT", "256"))
with torch.no_grad():
    _ffi._lib = libs[sys.argv[1]]
    xs = [torch.randn(1, 128, 184, 312, device="cuda:0") for _ in range(3)]
    acts = [c8.pack(x) for x in xs]
    layer = torch.nn.Conv2d(384, cout, 3, padding=1).cuda()
    cz, cr = (torch.randn(1, 128, 184, 312, device="cuda:0") for _ in range(2))
    rh = c8.ActC8(1, 128, 184, 312, "cuda:0")
    res = {k: [] for k in libs}
    for rnd in range(4):
        for k, L in libs.items():
            _ffi._lib = L
            if os.environ.get("C8_GATE"):       # the z|r layer with its gate epilogue (needs cout = 256)
                res[k].append(gtime(lambda: c8.gate_zr(acts, layer, cz, cr, xs[0], rh_c8=rh, cfg=cfg), 5, 6))
            else:
                res[k].append(gtime(lambda: c8.conv2d_c8(acts, layer, cfg=cfg), 5, 6))
    for k, v in res.items():
        print("%-50s min %.1f  median %.1f us   %s" % (os.path.basename(k), min(v), sorted(v)[len(v) // 2], " ".join("%.1f" % x for x in v)))
