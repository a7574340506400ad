"""bench.py with other settings of the loop (loop_c8): lower-case keys are tile shapes of loop_c8._CFG, upper-case keys module
attributes, e.g.
    python tools/bench_cfg.py zr16=3,q16=3,HEAD_FIRST=0 --skip-cpu-baseline --steps 15"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_stereo_amd.loop_c8 as lp  # noqa: E402

if len(sys.argv) > 1 and "=" in sys.argv[1]:
    for k, v in (kv.split("=") for kv in sys.argv[1].split(",")):
        if k.startswith("model."):
            import dkt_stereo_amd.raft_stereo as rs
            setattr(rs.RAFTStereo, k[6:], int(v))
        elif k.startswith("ex."):
            import dkt_stereo_amd.extractor as ex
            setattr(ex, k[3:], type(getattr(ex, k[3:]))(int(v)))
        elif k.isupper():
            setattr(lp, k, type(getattr(lp, k))(int(v)))
        else:
            lp._CFG[k] = int(v)
    sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
else:
    sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
