"""Times the 384->256 (cfg 1) and 384->128 (cfg 2) C8S convolutions at 184x312 for the library DKT_LIB_PATH points to."""
import os, sys, torch
os.environ.setdefault("DKT_ALLOW_ABLATION", "1")     # this tool times ablation builds (DKT_LIB_PATH)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from c8_check import gtime
from dkt_stereo_amd import conv_c8 as c8
torch.manual_seed(0)
with torch.no_grad():
    xs = [torch.randn(1, 128, 184, 312, device="cuda:0") for _ in range(3)]
    acts = [c8.pack(x) for x in xs]
    l256 = torch.nn.Conv2d(384, 256, 3, padding=1).cuda()
    l128 = torch.nn.Conv2d(384, 128, 3, padding=1).cuda()
    r = []
    for cfg, layer in ((1, l256), (2, l128), (3, l128)):
        r.append("cfg%d->%d %.1f" % (cfg, layer.weight.shape[0], gtime(lambda: c8.conv2d_c8(acts, layer, cfg=cfg), 5, 8)))
    print(os.environ.get("TAG", ""), " | ".join(r), flush=True)
