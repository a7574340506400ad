"""What the reference's multi-GPU entry point (nn.DataParallel, tools/ft_dkt.py:119-125; teachers called with test_mode=True,
:193,199) gets from this library at the benchmark shape: ms per 736x1248 pair, 32 iterations, for
  (a) the master module called directly (tools/evaluate_stereo.py:361 with device_ids=[0] does exactly this),
  (b) a fresh replica on a fresh thread per forward, as DataParallel makes them -- served by the persistent per-device copy
      (raft_stereo._Shadow: captured loop and packed weights survive),
  (c) the same with replica_shadows = False: the replica runs the plain un-captured loop itself (rounds 2-3 behaviour),
  (d) nn.DataParallel itself over device_ids=[0, 0] if torch accepts one device twice (two chunks of a batch of two).
    python tools/dp_replica_bench.py [N=10]"""
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _synth  # noqa: E402
from dkt_stereo_amd.raft_stereo import RAFTStereo  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = "cuda:0"
model = RAFTStereo()
model.load_state_dict(_synth.torch_state_dict(_synth.shapes_of(model), 7), strict=True)
model.to(dev).eval()
a, b = _synth.image_pair(1000, 1, 736, 1248, 12)
i1, i2 = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)


def via_replica():
    rep = model._replicate_for_data_parallel()
    out = []
    th = threading.Thread(target=lambda: out.append(rep(i1, i2, iters=32, test_mode=True)[1]))
    th.start()
    th.join()
    return out[0]


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3, r


with torch.no_grad():
    t_a, want = timed(lambda: model(i1, i2, iters=32, test_mode=True)[1], N)
    want = want.clone()
    t_b, got = timed(via_replica, N)
    same_b = torch.equal(got, want)
    model.replica_shadows = False
    t_c, got = timed(via_replica, max(2, N // 3))
    err_c = float((got - want).abs().max())
    model.replica_shadows = True
    print("(a) master called directly               %7.2f ms per pair" % t_a)
    print("(b) fresh replica + thread, shadow copy   %7.2f ms per pair   bit-identical to (a): %s" % (t_b, same_b))
    print("(c) fresh replica + thread, plain loop    %7.2f ms per pair   max |d| vs (a): %.2e" % (t_c, err_c))
    try:
        dp = torch.nn.DataParallel(model, device_ids=[0, 0])
        j1, j2 = torch.cat([i1, i1]), torch.cat([i2, i2])
        t_d, got = timed(lambda: dp(j1, j2, iters=32, test_mode=True)[1], max(2, N // 2))
        print("(d) nn.DataParallel(device_ids=[0, 0]), batch of two: %7.2f ms per pair   both chunks bit-identical to (a): %s"
              % (t_d / 2, bool(torch.equal(got[0:1], want) and torch.equal(got[1:2], want))))
    except Exception as e:        # noqa: BLE001
        print("(d) nn.DataParallel(device_ids=[0, 0]) not accepted by torch: %s" % (str(e).splitlines()[0],))
