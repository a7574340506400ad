"""Phase timeline of the fused ConvGRU kernel: builds a tools-side copy of csrc/gru_c8.hip whose `// @trace(k)` markers
store s_memrealtime (100 MHz) per block behind the error word, runs the cfg2 gru08 (+ gru32) launch and prints the
per-phase durations (median / min / max over blocks).  Optional text substitutions make timing-only variants
(`--sub 'gate_A();=>'`): the product source carries no ablation branches."""
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
OUT = os.path.join(HERE, "_build")


TAG = ([a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--tag=")] or [""])[0]


def build(subs, trace=True):
    from dkt_stereo_amd import build as B
    os.makedirs(OUT, exist_ok=True)
    src = open(os.path.join(B.CSRC, "gru_c8.hip")).read()
    if trace:
        import re
        # one record of 8 time stamps per (tile round of the block, block); later tiles of a block restart at stamp 0 behind
        # the previous tile's gate B (their "prologue" = the context terms + first fragments)
        src = re.sub(r"// @trace\((\d+)\)",
                     r"if (tid == 0) ((unsigned long long *)(ap.err + 16))[(trace_round * gridDim.x + blockIdx.x) * 8 + \1] = __builtin_amdgcn_s_memrealtime();", src)
        a = "    const int tid = threadIdx.x;\n"
        assert a in src
        src = src.replace(a, a + "    int trace_round = 0;\n", 1)
        a = "        tile = tn; b = nb; txy = ntxy; h0 = nh0; w0 = nw0;\n"
        assert a in src
        src = src.replace(a, a + "        ++trace_round;\n        if (tid == 0) ((unsigned long long *)(ap.err + 16))[(trace_round * gridDim.x + blockIdx.x) * 8 + 0] = __builtin_amdgcn_s_memrealtime();\n", 1)
        a = "        init_A(b, h0, w0);\n        G8_FIRST_FRAGS(2)\n    }\n"
        assert a in src
        src = src.replace(a, "        init_A(b, h0, w0);\n        G8_FIRST_FRAGS(2)\n        if (tid == 0) ((unsigned long long *)(ap.err + 16))[(trace_round * gridDim.x + blockIdx.x) * 8 + 1] = __builtin_amdgcn_s_memrealtime();\n    }\n", 1)
    for sub in subs:
        a, b = sub.split("=>")
        assert a in src, a
        src = src.replace(a, b)
    src = src.replace('#include "dkt_common.h"', '#include "%s/dkt_common.h"' % B.CSRC)
    path = os.path.join(OUT, "gru_c8_trace%s.hip" % TAG)
    open(path, "w").write(src)
    obj = os.path.join(OUT, "gru_c8_trace%s.o" % TAG)
    subprocess.check_call([B.HIPCC] + B.CFLAGS + B.EXTRA_FLAGS["gru_c8"] + ["-c", path, "-o", obj])
    objs = [os.path.join(B.OBJ_DIR, f) for f in sorted(os.listdir(B.OBJ_DIR)) if f.endswith(".o") and f != "gru_c8.o"]
    lib = os.path.join(OUT, "libdktstereo_trace%s.so" % TAG)
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", obj] + objs + ["-o", lib])
    return lib


def run(lib, pair=True, reps=300, B=1):
    os.environ["DKT_LIB_PATH"] = lib
    from dkt_stereo_amd import conv_c8 as c8
    sys.path.insert(0, HERE)
    import gru_c8_check as chk
    big = chk.State(*chk.make(B, 184, 312, [128, 128], 1))
    small = chk.State(*chk.make(B, 46, 78, [128], 2))          # the loop's rider: gru32 at 1/16 resolution
    rounds = 12
    err = torch.zeros(16 + 2 * 8 * 256 * rounds, device="cuda", dtype=torch.int32)
    passes = int(([a.split("=")[1] for a in sys.argv[1:] if a.startswith("--passes=")] or ["3"])[0])
    with c8.passes(passes):              # MFMA products per block of the traced launch (1 = what args.mixed_precision runs)
        d0, d1 = big.desc(), small.desc()
    print("passes %d" % passes)
    reps = max(3, reps // B)
    for _ in range(reps):
        err[16:].zero_()
        c8.gru_launch(d0, d1 if pair else None, err=err)
    torch.cuda.synchronize()
    t_all = err[16:].view(torch.int64).view(rounds, 256, 8).cpu().double() * 0.01       # us; [round][block][stamp]
    nblk = int((t_all[0, :, 0] > 0).sum())
    names = ["prologue", "phase A", "gate A + publish", "phase B", "gate B"]
    t0 = t_all[0, :nblk, 0].min()
    last = t_all[:, :, 5].max()
    print("batch %d: blocks %d; kernel span %.1f us (first start -> last end)" % (B, nblk, float(last - t0)))
    print("start skew: %.1f us" % float(t_all[0, :nblk, 0].max() - t0))
    # blocks of the finest level (problem 0) only: the rider's blocks are the launch's last ones
    tiles0 = B * 230
    nb0 = nblk if not pair else (nblk - min(18 * B, max(1, round(nblk * (18 * B * 32.0) / (tiles0 * 48.0 + 18 * B * 32.0)))) if tiles0 + 18 * B > nblk else tiles0)
    for r in range(rounds):
        t = t_all[r, :nb0]
        t = t[(t[:, 0] > 0) & (t[:, 5] > 0)]
        if t.shape[0] == 0:
            break
        print("round %d: %d tiles, start at median %.1f us, done at median %.1f (max %.1f)" %
              (r, t.shape[0], float((t[:, 0] - t0).median()), float((t[:, 5] - t0).median()), float((t[:, 5] - t0).max())))
        for k, n in enumerate(names):
            d = t[:, k + 1] - t[:, k]
            print("  %-18s median %.1f  min %.1f  max %.1f us" % (n, float(d.median()), float(d.min()), float(d.max())))
        if r == 0 and "--by-xcd" in sys.argv:
            # which blocks are the slow ones: phase A per XCD (block index & 7) and for the riders' neighbours
            ta = t_all[0, :nb0]
            for x in range(8):
                d = (ta[x::8, 2] - ta[x::8, 1])
                d = d[d > 0]
                print("    XCD %d: %3d blocks, phase A median %.1f  max %.1f; done median %.1f" %
                      (x, d.numel(), float(d.median()), float(d.max()), float((ta[x::8, 5] - t0)[ta[x::8, 5] > 0].median())))
        if r == 0:
            for k, n in ((2, "end of phase A"), (3, "published"), (4, "end of phase B"), (5, "done")):
                d = t[:, k] - t0
                print("  %-18s at median %.1f  min %.1f  max %.1f us" % (n, float(d.median()), float(d.min()), float(d.max())))
    print("err word", int(err[0].item()))


if __name__ == "__main__":
    subs = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--sub=")]
    if "--build-only" in sys.argv:
        print(build(subs))
    else:
        lib = os.path.join(OUT, "libdktstereo_trace%s.so" % TAG)
        if not os.path.exists(lib) or "--rebuild" in sys.argv:
            lib = build(subs)
        bs = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--batch=")] or [1]
        for B_ in bs:
            run(lib, pair="--single" not in sys.argv, B=B_)
