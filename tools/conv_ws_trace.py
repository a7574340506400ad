"""Per-tile timeline of the weights-stationary convolution: builds a tools-side copy of csrc/conv2d.hip + conv_ws.h whose
`// @trace(k)` markers store s_memtime (shader cycles) per wave and tile, runs one 736 x 1248 launch of each epilogue form and
prints the cycles per chunk (MFMA floor: 108 MFMAs x 32 = 3456), per epilogue and per tile.  The product source carries only
the comment markers.   python tools/conv_ws_trace.py [--sub 'a=>b' ...]"""
import ctypes
import os
import re
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
OUT = os.path.join(HERE, "_build")
NT = 16          # tiles traced per wave


def build(subs):
    from dkt_stereo_amd import build as B
    os.makedirs(OUT, exist_ok=True)
    ws = open(os.path.join(B.CSRC, "conv_ws.h")).read()
    ws = ws.replace("template <int NRM, int EPI>\n__global__",
                    "__device__ unsigned long long *ws_trace_buf = nullptr;\n"
                    "extern \"C\" int dkt_ws_trace_set(void *p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(ws_trace_buf), &p, sizeof(p)); }\n"
                    "template <int NRM, int EPI>\n__global__", 1)
    ws = ws.replace("    int tile = lb;\n", "    int tile = lb;\n    int tcount = 0;\n", 1)
    ws = re.sub(r"// @trace\(([^)]*)\)",
                r"if (ws_trace_buf && lane == 0 && tcount < %d) ws_trace_buf[(((long)blockIdx.x * 4 + wave) * %d + tcount) * 8 + (\1)] = __builtin_amdgcn_s_memtime();" % (NT, NT), ws)
    ws = ws.replace("        if (!have_next) break;\n", "        ++tcount;\n        if (!have_next) break;\n", 1)
    for sub in subs:
        a, b = sub.split("=>")
        assert a in ws, a
        ws = ws.replace(a, b)
    open(os.path.join(OUT, "conv_ws.h"), "w").write(ws)
    src = open(os.path.join(B.CSRC, "conv2d.hip")).read()
    src = src.replace('#include "dkt_common.h"', '#include "%s/dkt_common.h"' % B.CSRC)
    src = src.replace('#include "conv_ws.h"', '#include "%s/conv_ws.h"' % OUT)
    path = os.path.join(OUT, "conv2d_ws_trace.hip")
    open(path, "w").write(src)
    obj = os.path.join(OUT, "conv2d_ws_trace.o")
    subprocess.check_call([B.HIPCC] + B.CFLAGS + ["-DCONV_TU_PASSES=3", "-c", path, "-o", obj])
    objs = [os.path.join(B.OBJ_DIR, f) for f in sorted(os.listdir(B.OBJ_DIR)) if f.endswith(".o") and f != "conv2d_tu3.o"]
    lib = os.path.join(OUT, "libdktstereo_wstrace.so")
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", obj] + objs + ["-o", lib])
    return lib


def run(lib):
    os.environ["DKT_LIB_PATH"] = lib
    import torch.nn as nn
    from dkt_stereo_amd import _ffi, conv
    from dkt_stereo_amd.extractor import instance_norm_params
    L = _ffi.lib()
    L.dkt_ws_trace_set.argtypes = [ctypes.c_void_p]
    L.dkt_ws_trace_set.restype = ctypes.c_int
    layer = nn.Conv2d(64, 64, 3, padding=1).cuda()
    x = torch.randn(1, 64, 736, 1248, device="cuda")
    res = torch.randn(1, 64, 736, 1248, device="cuda").relu()
    p = instance_norm_params(nn.InstanceNorm2d(64), x)
    forms = {"relu": lambda: conv.conv2d(x, layer, relu=True),
             "join": lambda: conv.conv2d_fused(x, layer, relu=True, residual=res),
             "in_norm+stats": lambda: conv.conv2d_stats(x, layer, in_norm=p)}
    buf = torch.zeros(256 * 4 * NT * 8, device="cuda", dtype=torch.int64)
    with torch.no_grad():
        for name, f in forms.items():
            assert L.dkt_ws_trace_set(None) == 0
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            buf.zero_()
            assert L.dkt_ws_trace_set(ctypes.c_void_p(buf.data_ptr())) == 0
            f()
            torch.cuda.synchronize()
            t = buf.view(256, 4, NT, 8).cpu().double()
            ok = t[..., 5] > 0
            full = ok[:, :, :13].all(dim=2)                                  # waves with at least 13 traced tiles
            tt = t[full][:, 1:13]                                            # steady state: tiles 1 .. 12
            chunks = [tt[..., 1] - tt[..., 0]] + [tt[..., k + 1] - tt[..., k] for k in range(1, 4)]
            epi = tt[..., 5] - tt[..., 4]
            tile = tt[..., 5] - tt[..., 0]
            gap = t[full][:, 2:14, 0] - t[full][:, 1:13, 5]                  # epilogue end -> next tile's first marker
            print("%-14s waves %d;  per tile %.0f ticks = chunks %s (108 MFMAs = 3456 cycles each) + epilogue %.0f + %.0f between tiles"
                  % (name, int(full.sum()), float(tile.mean() + gap.mean()),
                     " ".join("%.0f" % float(c.mean()) for c in chunks), float(epi.mean()), float(gap.mean())), flush=True)


if __name__ == "__main__":
    subs = [sys.argv[i + 1] for i, a in enumerate(sys.argv) if a == "--sub"]
    # --run-only: the library was built on the development host (the object directory does not travel to the GPU box)
    run(os.path.join(OUT, "libdktstereo_wstrace.so") if "--run-only" in sys.argv else build(subs))
