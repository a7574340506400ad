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
