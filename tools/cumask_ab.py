#!/usr/bin/env python3
"""A/B of a CU mask on the forked stream of the refinement loop: the middle GRU's chain (resample -> gru16
z|r -> q -> resample) on a stream created with hipExtStreamCreateWithCUMask over the first N CUs (bit i = CU i; the driver deals
mask bits round-robin over the 8 XCDs, so a multiple of 8 takes the same number of CUs from every XCD), the main chain (flow
head -> motion front -> convc2 | convf2 -> encoder.conv) unmasked -- so that the front's blocks always find free CUs instead of
waiting for the side chain's resident blocks to retire.  Captured graphs do not carry a stream's CU mask into their kernel
nodes, so the comparison runs the loop's units as plain launches (RAFTStereo.c8_eager) in every arm, mask or not.

    python tools/cumask_ab.py [N ...]      (default: 0 = no mask, 224 192 160 128; negative: -N bits of every 32; two alternations)
Prints ms per iteration of the loop and the in-pipeline duration of the motion front per arm."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _synth  # noqa: E402

DEV = torch.device("cuda", 0)


def masked_stream(n_cus, total=256):
    """n_cus > 0: the first n_cus mask bits.  n_cus < 0: -n_cus bits of EVERY 32-bit word -- the same number of CUs from every
    XCD whether the driver lays the mask out XCD-major (bit = 32 * xcd + cu) or deals it round-robin (bit = 8 * cu + xcd)."""
    hip = ctypes.CDLL("libamdhip64.so")
    words = (total + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    if n_cus < 0:
        for w in range(words):
            mask[w] = (1 << (-n_cus)) - 1
    for i in range(max(n_cus, 0)):
        mask[i // 32] |= 1 << (i % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), words, mask)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask failed: %d" % rc)
    return torch.cuda.ExternalStream(st.value, device=DEV)


@torch.no_grad()
def main():
    from dkt_stereo_amd import conv_c8 as dc8
    from dkt_stereo_amd import update as upd
    from dkt_stereo_amd.raft_stereo import RAFTStereo
    arms = [int(a) for a in sys.argv[1:]] or [0, 224, 192, 160, 128]
    torch.cuda.set_device(DEV)
    model = RAFTStereo()
    model.load_state_dict(_synth.torch_state_dict(_synth.shapes_of(model), 7))
    model.to(DEV).eval()
    i1, i2 = (torch.from_numpy(t).to(DEV) for t in _synth.image_pair(1000, 1, 736, 1248, 12))
    for _ in range(3):
        model(i1, i2, iters=32, test_mode=True)
    torch.cuda.synchronize()
    model.c8_eager = True
    model.check_finite = False
    plain = upd._side_stream(DEV, slot=0)
    streams = {0: plain}
    for n in arms:
        if n and n not in streams:
            streams[n] = masked_stream(n)
    key = (DEV.index, 0)
    real_front = dc8.motion_front
    for rnd in range(2):
        for n in arms:
            upd._SIDE_STREAMS.streams[key] = streams[n]
            fr = []

            def timed_front(*a, **k):
                ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ea.record()
                real_front(*a, **k)
                eb.record()
                fr.append((ea, eb))

            ms = []
            for rep in range(4):
                fm = model.encode(i1, i2)
                torch.cuda.synchronize()
                dc8.motion_front = timed_front if rep == 3 else real_front
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                model.iterate(*fm, 32)
                e1.record()
                torch.cuda.synchronize()
                dc8.motion_front = real_front
                if rep:
                    ms.append(e0.elapsed_time(e1))
            front_us = 1e3 * sum(a.elapsed_time(b) for a, b in fr) / max(len(fr), 1)
            label = ("%d of every 32" % -n) if n < 0 else (n or "all")
            print("round %d  side stream on %s CUs: loop %.3f ms per iteration (%.2f / %.2f / %.2f ms per pair), motion front %.1f us in "
                  "the pipeline" % (rnd, label, sum(ms) / len(ms) / 32, ms[0], ms[1], ms[2], front_us), flush=True)
    upd._SIDE_STREAMS.streams[key] = plain


if __name__ == "__main__":
    main()
