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


def build(subs, trace=True):
    from dkt_stereo_amd import build as B
    os.makedirs(OUT, exist_ok=True)
    src = open(os.path.join(B.CSRC, "gru_c8.hip")).read()
    if trace:
        import re
        src = re.sub(r"// @trace\((\d+)\)",
                     r"if (tid == 0) ((unsigned long long *)(ap.err + 16))[blockIdx.x * 8 + \1] = __builtin_amdgcn_s_memrealtime();", src)
    for sub in subs:
        a, b = sub.split("=>")
        assert a in src, a
        src = src.replace(a, b)
    src = src.replace('#include "dkt_common.h"', '#include "%s/dkt_common.h"' % B.CSRC)
    path = os.path.join(OUT, "gru_c8_trace.hip")
    open(path, "w").write(src)
    obj = os.path.join(OUT, "gru_c8_trace.o")
    subprocess.check_call([B.HIPCC] + B.CFLAGS + B.EXTRA_FLAGS["gru_c8"] + ["-c", path, "-o", obj])
    objs = [os.path.join(B.OBJ_DIR, f) for f in sorted(os.listdir(B.OBJ_DIR)) if f.endswith(".o") and f != "gru_c8.o"]
    lib = os.path.join(OUT, "libdktstereo_trace.so")
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", obj] + objs + ["-o", lib])
    return lib


def run(lib, pair=True, reps=300):
    os.environ["DKT_LIB_PATH"] = lib
    from dkt_stereo_amd import conv_c8 as c8
    sys.path.insert(0, HERE)
    import gru_c8_check as chk
    big = chk.State(*chk.make(1, 184, 312, [128, 128], 1))
    small = chk.State(*chk.make(1, 23, 39, [128], 2))
    err = torch.zeros(16 + 2 * 8 * 300, device="cuda", dtype=torch.int32)
    d0, d1 = big.desc(), small.desc()
    for _ in range(reps):
        c8.gru_launch(d0, d1 if pair else None, err=err)
    torch.cuda.synchronize()
    t = err[16:].view(torch.int64).view(-1, 8)[:248 if pair else 230].cpu().double() * 0.01       # us
    names = ["prologue", "phase A", "gate A + publish", "phase B", "gate B"]
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    print("blocks %d; kernel span %.1f us (first start -> last end)" % (t.shape[0], float(t[:, 5].max() - t0)))
    print("start skew: %.1f us" % float(t[:, 0].max() - t0))
    for k, n in enumerate(names):
        d = t[:230, k + 1] - t[:230, k]
        print("  %-18s median %.1f  min %.1f  max %.1f us" % (n, float(d.median()), float(d.min()), float(d.max())))
    for k, n in ((2, "end of phase A"), (3, "published"), (4, "end of phase B"), (5, "done")):
        d = t[:230, k] - t0
        print("  %-18s at median %.1f  min %.1f  max %.1f us" % (n, float(d.median()), float(d.min()), float(d.max())))
    print("err word", int(err[0].item()))


if __name__ == "__main__":
    subs = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--sub=")]
    if "--build-only" in sys.argv:
        print(build(subs))
    else:
        lib = os.path.join(OUT, "libdktstereo_trace.so")
        if not os.path.exists(lib) or "--rebuild" in sys.argv:
            lib = build(subs)
        run(lib, pair="--single" not in sys.argv)
