"""C8S loop vs the round-2 loop on the same model / pair: final disparity difference and time per pair."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _synth
from dkt_stereo_amd.raft_stereo import RAFTStereo
DEV = "cuda:0"
with torch.no_grad():
    sizes = ((736, 1248, 32),) if os.environ.get("C8_ONLY") else ((256, 512, 8), (736, 1248, 32))
    for (Hh, Ww, it) in sizes:
        m = RAFTStereo()
        m.load_state_dict(_synth.torch_state_dict(_synth.shapes_of(m), 7))
        m.to(DEV).eval()
        i1, i2 = _synth.image_pair(3, 1, Hh, Ww, 40)
        i1, i2 = torch.from_numpy(i1).to(DEV), torch.from_numpy(i2).to(DEV)
        res = {}
        for c8 in ((True,) if os.environ.get("C8_ONLY") == "1" else (False,) if os.environ.get("C8_ONLY") == "0" else (False, True)):
            m.use_c8 = c8
            m._graph_state = None if hasattr(m, "_graph_state") else None
            for _ in range(3):
                _, up = m(i1, i2, iters=it, test_mode=True)
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(5):
                _, up = m(i1, i2, iters=it, test_mode=True)
            torch.cuda.synchronize()
            res[c8] = (up.clone(), (time.time() - t0) / 5 * 1e3)
        if len(res) < 2:
            print("ms/pair", {k: v[1] for k, v in res.items()}); continue
        d = (res[True][0] - res[False][0]).abs()
        print("%dx%d it%d: c8 vs round-2 max|d| %.2e mean %.2e ; ms/pair round-2 %.2f c8 %.2f" % (Hh, Ww, it, float(d.max()), float(d.mean()), res[False][1], res[True][1]), flush=True)
