import os, sys, time, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import _synth
import dkt_stereo_amd.update as upd
from dkt_stereo_amd.raft_stereo import RAFTStereo
m = RAFTStereo(); m.load_state_dict(_synth.torch_state_dict(_synth.shapes_of(m), 7)); m.cuda().eval()
i1, i2 = _synth.image_pair(1000, 1, 736, 1248, 12)
i1, i2 = torch.from_numpy(i1).cuda(), torch.from_numpy(i2).cuda()
with torch.no_grad():
    fm = m.encode(i1, i2)
    for fuse in (True, False, True, False):
        upd.FUSE_GATES = fuse
        m._graph_state = None
        for _ in range(2): m.iterate(*fm, 32)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): m.iterate(*fm, 32)
        torch.cuda.synchronize()
        print("fuse_gates=%s hot path %.2f ms" % (fuse, (time.perf_counter() - t0) / 3 * 1e3), flush=True)
    # direct kernels alone
    sys.path.insert(0, 'tools')
    import bench_kernels as bk
    from dkt_stereo_amd import conv
    for (cin, cout, k, relu) in ((256, 2, 3, False), (2, 64, 7, True)):
        layer = torch.nn.Conv2d(cin, cout, k, padding=k // 2).cuda()
        x = torch.randn(1, cin, 184, 312, device="cuda")
        for be in ("f16x3", "miopen"):
            conv.set_backend(be)
            bk.report("conv %d->%d k%d %s" % (cin, cout, k, be), bk.timeit(lambda: conv.conv2d(x, layer, relu=relu), n=30, warm=3))
    conv.set_backend("f16x3")
