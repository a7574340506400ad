"""bench.py with other tile shapes for the loop's layer classes (loop_c8._CFG), e.g.
    python tools/bench_cfg.py zr16=3,q16=3 --skip-cpu-baseline --steps 15"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dkt_stereo_amd.loop_c8 as lp  # noqa: E402

if len(sys.argv) > 1 and "=" in sys.argv[1]:
    lp._CFG.update({k: int(v) for k, v in (kv.split("=") for kv in sys.argv[1].split(","))})
    sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
else:
    sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
