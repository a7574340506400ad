"""Stress test for the few-output 3x3 kernel's LDS-DMA staging beside another kernel that holds LDS on the same CU:
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
        for it in range(100):
            g.replay()
            torch.cuda.synchronize()
            bad_f += sum(0 if torch.equal(y, ref) else 1 for y in yf)
            bad_s += sum(0 if torch.equal(y, ref_s) else 1 for y in ys)
        print("cout %d: few-kernel results wrong %d/1200, co-resident conv results wrong %d/600" % (cout, bad_f, bad_s), flush=True)
