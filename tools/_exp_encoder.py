"""Encoder timing: fnet (both images) and cnet alone, and encode() with / without the two-stream overlap."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _synth
from dkt_stereo_amd.raft_stereo import RAFTStereo
DEV = "cuda:0"
def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
with torch.no_grad():
    m = RAFTStereo(); m.load_state_dict(_synth.torch_state_dict(_synth.shapes_of(m), 7)); m.to(DEV).eval()
    i1, i2 = [torch.from_numpy(a).to(DEV) for a in _synth.image_pair(2000, 1, 736, 1248, 12)]
    a1 = (2 * (i1 / 255.0) - 1.0).contiguous(); a2 = (2 * (i2 / 255.0) - 1.0).contiguous()
    print("fnet [2 images]: %.2f ms" % timeit(lambda: m.fnet([a1, a2])))
    print("cnet [1 image] : %.2f ms" % timeit(lambda: m.cnet(a1, num_layers=3)))
    print("encode (2 streams): %.2f ms" % timeit(lambda: m.encode(i1, i2)))
    m.encoder_streams = False
    print("encode (1 stream) : %.2f ms" % timeit(lambda: m.encode(i1, i2)))
    # layer-level: fnet pieces
    x = torch.cat([a1, a2], 0)
    f = m.fnet
    from dkt_stereo_amd.extractor import conv_norm_act
    t0 = timeit(lambda: conv_norm_act(f.conv1, f.norm1, x, True))
    y = conv_norm_act(f.conv1, f.norm1, x, True)
    t1 = timeit(lambda: f.layer1(y)); y1 = f.layer1(y)
    t2 = timeit(lambda: f.layer2(y1)); y2 = f.layer2(y1)
    t3 = timeit(lambda: f.layer3(y2)); y3 = f.layer3(y2)
    t4 = timeit(lambda: f.conv2(y3))
    print("fnet: stem %.2f  layer1 %.2f  layer2 %.2f  layer3 %.2f  conv2 %.2f ms" % (t0, t1, t2, t3, t4))
