"""Why are encoder convs 3.4x slower inside the pipeline than standalone?  Replays a captured
pipeline input (data effect) and rotates cold buffers (cache effect)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _synth
from dkt_stereo_amd import conv
from dkt_stereo_amd.raft_stereo import RAFTStereo
DEV = "cuda:0"
def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n
with torch.no_grad():
    m = RAFTStereo(); m.load_state_dict(_synth.torch_state_dict(_synth.shapes_of(m), 7)); m.to(DEV).eval()
    i1, i2 = [torch.from_numpy(a).to(DEV) for a in _synth.image_pair(2000, 1, 736, 1248, 12)]
    cap = {}
    orig = conv.conv2d
    seen = set()
    def spy(x, layer, relu=False, out=None):
        sh = (tuple(x.shape) if not isinstance(x, (list, tuple)) else "list", tuple(layer.weight.shape))
        if sh not in seen:
            seen.add(sh); print("conv2d call", sh, flush=True)
        if not isinstance(x, (list, tuple)) and x.shape[1] == 64 and x.shape[2] == 368 and layer.weight.shape[0] == 64 and layer.weight.shape[2] == 3 and x.shape[0] not in cap:
            cap[x.shape[0]] = (x.clone(), layer)
        return orig(x, layer, relu, out)
    conv.conv2d = spy
    import dkt_stereo_amd.extractor as ex
    ex_conv = getattr(ex, "conv2d", None)
    if ex_conv is not None: ex.conv2d = spy
    m(i1, i2, iters=2, test_mode=True)
    conv.conv2d = orig
    if ex_conv is not None: ex.conv2d = orig
    print("captured", {k: tuple(v[0].shape) for k, v in cap.items()})
    for B, (x, layer) in cap.items():
        print("B=%d pipeline data: absmax %.3g mean|x| %.3g zeros %.2f  w absmax %.3g" % (B, x.abs().max(), x.abs().mean(), (x == 0).float().mean(), layer.weight.abs().max()))
        print("  replay captured input, same buffer: %.1f us" % timeit(lambda: conv.conv2d(x, layer)))
        r = torch.randn_like(x)
        print("  randn input, same layer:            %.1f us" % timeit(lambda: conv.conv2d(r, layer)))
        l2 = torch.nn.Conv2d(64, 64, 3, padding=1).to(DEV)
        print("  captured input, default-init layer: %.1f us" % timeit(lambda: conv.conv2d(x, l2)))
        bufs = [x.clone() for _ in range(6)]
        k = [0]
        def rot():
            k[0] = (k[0] + 1) % len(bufs)
            return conv.conv2d(bufs[k[0]], layer)
        print("  rotating 6 cold buffers:            %.1f us" % timeit(rot, n=24))
        xs = x * 1e-4
        print("  captured*1e-4 (fp16 subnormal range): %.1f us" % timeit(lambda: conv.conv2d(xs, layer)))
