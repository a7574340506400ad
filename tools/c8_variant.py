#!/usr/bin/env python3
"""Builds a tools-side variant of csrc/conv_c8.hip (text substitutions: `--sub=A=>B`, several allowed) into
tools/_build/libdktstereo_c8<tag>.so -- timing-only ablations; the product source carries no ablation branches.  Run the variant
with DKT_LIB_PATH=<that library> in front of any tool (tools/window_kernels.py, bench.py ...).
    python tools/c8_variant.py --tag=noepi '--sub=        epilogue();\n        publish(tile);=>        if (a.H < 0) epilogue();\n        publish(tile);'
`noepi` (what round 6 used for VERDICT r05 item 6): every tile's epilogue behind a condition that is never true (the accumulators stay live) -- no gates, no stores: the difference to the
product library is the most that ANY overlap of one work item's epilogue with the next item's MFMAs could hide."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
OUT = os.path.join(HERE, "_build")


def main():
    from dkt_stereo_amd import build as B
    B.build()
    tag = ([a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--tag=")] or ["variant"])[0]
    subs = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--sub=")]
    os.makedirs(OUT, exist_ok=True)
    src = open(os.path.join(B.CSRC, "conv_c8.hip")).read()
    for sub in subs:
        a, b = sub.encode().decode("unicode_escape").split("=>")
        assert a in src, a
        src = src.replace(a, b)
    for inc in ("dkt_common.h",):
        src = src.replace('#include "%s"' % inc, '#include "%s/%s"' % (B.CSRC, inc))
    path = os.path.join(OUT, "conv_c8_%s.hip" % tag)
    open(path, "w").write(src)
    obj = os.path.join(OUT, "conv_c8_%s.o" % tag)
    subprocess.check_call([B.HIPCC] + B.CFLAGS + B.EXTRA_FLAGS["conv_c8"] + ["-I", B.CSRC, "-c", path, "-o", obj])
    objs = [os.path.join(B.OBJ_DIR, f) for f in sorted(os.listdir(B.OBJ_DIR)) if f.endswith(".o") and f != "conv_c8.o"]
    lib = os.path.join(OUT, "libdktstereo_c8%s.so" % tag)
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", obj] + objs + ["-o", lib])
    print(lib)


if __name__ == "__main__":
    main()
