"""Stress test for the few-output 3x3 kernel beside another kernel that holds LDS on the same CU
(DKT_FEW_LDS_EXACT=1 lets them share a CU; DKT_LIB_PATH=dkt_stereo_amd/lib/variants/lib_<name>.so selects an instrumented
or differently compiled build made with tools/build_variant.sh, e.g. `fewdbg conv_direct -DFEW_DEBUG`):
a captured graph runs 12 launches of the 256 -> cout layer on one stream while a second stream runs the 1/16-resolution GRU
convolution (33 KB of LDS per block); every result is compared with a result computed alone.  (Before the kernel
claimed the CU's whole LDS, cout = 1 -- 123 KB -- was wrong in 1000 of 1200 launches here.)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dkt_stereo_amd import conv
torch.manual_seed(0)
dev = "cuda:0"
with torch.no_grad():
    for cout in (1, 2):
        H, W = 64, 128
        layer = torch.nn.Conv2d(256, cout, 3, padding=1).to(dev)
        small = torch.nn.Conv2d(384, 256, 3, padding=1).to(dev)        # gru16-like: 256 co x 1 row tiles (32.6 KB LDS)
        xs = torch.randn(1, 384, 32, 64, device=dev)
        x = torch.randn(1, 256, H, W, device=dev)
        ref = conv.conv2d(x, layer).clone()
        ref_s = conv.conv2d(xs, small, relu=True).clone()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        ys, yf = [], []
        with torch.cuda.graph(g):
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for _ in range(6):
                    ys.append(conv.conv2d(xs, small, relu=True))
            for _ in range(12):
                yf.append(conv.conv2d(x, layer))
            main.wait_stream(side)
        bad_f = bad_s = 0
        shown = 0
        for it in range(100):
            g.replay()
            torch.cuda.synchronize()
            bad_f += sum(0 if torch.equal(y, ref) else 1 for y in yf)
            for y in yf:
                if shown < 4 and not torch.equal(y, ref):
                    shown += 1
                    d = (y - ref)[0, 0]
                    nz = d.abs() > 0
                    rows = nz.any(1).nonzero().flatten().tolist()
                    cols = nz.any(0).nonzero().flatten().tolist()
                    print("  wrong pixels %d, rows %s, cols %d..%d (n=%d), max |d| %.3g" % (int(nz.sum()), rows[:12], cols[0], cols[-1], len(cols), float(d.abs().max())))
                    # per 4x64 tile: does the error equal (minus) a wave's partial sum, or a partial sum of another tile?
                    for th in range(0, H, 4):
                        for tw in range(0, W, 64):
                            dt = d[th:th + 4, tw:tw + 64]
                            if float(dt.abs().max()) == 0:
                                continue
                            msg = "  tile (%d,%d): %d wrong px" % (th, tw, int((dt.abs() > 0).sum()))
                            if shown == 1:
                                wc = (dt.abs() > 0).any(0).nonzero().flatten().tolist()
                                msg += " cols " + ",".join(str(c) for c in wc)
                            print(msg)
            bad_s += sum(0 if torch.equal(y, ref_s) else 1 for y in ys)
        try:
            import ctypes
            from dkt_stereo_amd import _ffi
            cnt = (ctypes.c_int * 16)()
            ctypes.CDLL(_ffi.LIB_PATH).dkt_debug_few_counters(cnt, 1)
            print("debug counters: dma-mismatch %d (first: alloc %08x wave %d chunk %d piece %d lane %d; landed elsewhere in the wave's buffers %d, nowhere %d) weights %d masked-nonzero %d blocks base!=0 %d / %d; compute-time reads differing from global %d (first: lane %d elems-mask %x wave %d alloc %08x)"
                  % (cnt[0], cnt[1] & 0xffffffff, cnt[2], cnt[3], cnt[4], cnt[5], cnt[11], cnt[10], cnt[6], cnt[7], cnt[8], cnt[9], cnt[12], cnt[13] >> 16, (cnt[13] >> 4) & 0xfff, cnt[13] & 15, cnt[14] & 0xffffffff))
        except AttributeError:
            pass
        print("exact-LDS=%s " % os.environ.get("DKT_FEW_LDS_EXACT", "0") + "cout %d: few-kernel results wrong %d/1200, co-resident conv results wrong %d/600" % (cout, bad_f, bad_s), flush=True)
