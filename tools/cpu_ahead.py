"""How far the host runs ahead of the device: host time per forward (no synchronisation) against device time per forward."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _synth
from dkt_stereo_amd.raft_stereo import RAFTStereo
dev = torch.device("cuda", 0)
model = RAFTStereo()
model.load_state_dict(_synth.torch_state_dict(_synth.shapes_of(model), 7), strict=True)
model.to(dev).eval()
i1, i2 = (torch.from_numpy(t).to(dev) for t in _synth.image_pair(1000, 1, 736, 1248, 12))
with torch.no_grad():
    for _ in range(6):
        model(i1, i2, iters=32, test_mode=True)
    torch.cuda.synchronize()
    model.check_finite = False
    for n in (1, 5, 20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            model(i1, i2, iters=32, test_mode=True)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("n=%d: host %.2f ms per forward (enqueue only), total %.2f ms per forward" % (n, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        model(i1, i2, iters=32, test_mode=True)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
