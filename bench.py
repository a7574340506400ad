#!/usr/bin/env python3
"""bench.py -- stereo pairs/s of the RAFT-Stereo inference path on MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): RAFT-Stereo, 736x1248 KITTI-shape synthetic
pair (D=192 range), 32 GRU iterations, batch 1 per GPU, fp32, random-init
weights of the reference architecture, inputs resident in HBM before the timed
region.  One "step" = one whole test_mode forward of one batch: both encoders,
correlation-volume build, 32 x (pyramid lookup + update block), convex
upsampling.  With N GPUs every rank runs its own pairs (weak scaling, no
data-path collective); the final disparity maps are gathered to rank 0 over
RCCL inside the timed region, as a real sharded evaluation would.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      the dominant kernel (update-block convolution, MFMA-bound): algorithmic flops
                per launch / its average launch duration measured live with HIP events on the
                launch stream (frac = frac_algorithmic; mfma_issue_frac counts the 3 fp16 MFMA
                passes per product); roofline_lookup: the corr-lookup kernel (HBM-bound, the
                kernel BASELINE.json's north_star sets the 60 % target for) -- since round 2 it
                is fused with the 1x1 layer that consumes it; stand-alone launch time, plus
                `in_pipeline_*`: the same kernel as the loop runs it (rocprofv3 trace of this
                command, profiles/r03_pair_breakdown.txt).  `traffic`: 2 x FETCH_SIZE +
                WRITE_SIZE from profiles/r03_hbm_traffic.txt (separate rocprofv3 --pmc passes
                of the same kernels and shapes, calibrated on known-traffic streams; the
                round-2 kernels' figures are in profiles/r02_hbm_traffic.txt)
  cpu_baseline  oracle/torch_oracle.py (pure-PyTorch CPU port of the reference,
                pinned to it by tests/golden) timed on this host, bounded sample
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
FP32_MFMA_PEAK_TFLOPS = 157.3
FP16_MFMA_PEAK_TFLOPS = 2500.0  # dense


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--height", type=int, default=736)
    p.add_argument("--width", type=int, default=1248)
    p.add_argument("--iters", type=int, default=32)
    p.add_argument("--batch", type=int, default=None,
                   help="pairs per GPU per step (default 1; with --gpus > 1: 8 = BASELINE.json configs[3], batch 64 over 8 GPUs)")
    p.add_argument("--cpu-iters", type=int, default=6, help="GRU iterations timed by the CPU baseline sample")
    p.add_argument("--skip-cpu-baseline", action="store_true")
    p.add_argument("--pmc", action="store_true",
                   help="measure roofline.traffic in this run: two rocprofv3 --pmc sub-runs (FETCH_SIZE, WRITE_SIZE) of the reported launches")
    p.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                   help="torch.distributed backend for --gpus > 1: nccl = RCCL over xGMI (default); gloo = host-staged, lets several "
                        "ranks share one device (tests/test_gpu_round5.py exercises the N > 1 branch that way on a 1-GPU box)")
    p.add_argument("--schedule", default=None, metavar="K1,K2",
                   help="precision schedule of the refinement loop (loop_c8.SCHEDULE): the first K1 iterations at one fp16 MFMA product "
                        "per block, the next K2 at two, the rest fp32-class.  A SEPARATE line, never the headline: `dtype` says what ran")
    p.add_argument("--distinct-pairs", type=int, default=8, metavar="N",
                   help="after the timed region: N DIFFERENT seeded pairs (shifts 12 / 40, amplitudes x0.5 ... x2) through the same model in "
                        "the product's default mode (check_finite = True), each step timed on its own; reports recalibrations and the "
                        "worst step (0 = skip)")
    p.add_argument("--mixed-precision", action="store_true",
                   help="args.mixed_precision = True (raft_stereo.py:95,156; the DKT teachers' default, tools/ft_dkt.py:317): encoders and "
                        "refinement loop at one fp16 MFMA product per block.  A SEPARATE line, never the headline")
    p.add_argument("--conv-backend", default=None, choices=["f16x3", "f16x2", "f16", "miopen"],
                   help="update-block convolution path (default: the package default, f16x3)")
    a = p.parse_args()
    if a.batch is None:
        a.batch = 8 if a.gpus > 1 else 1
    return a


def lookup_bytes_per_launch(n_pixels, L=4, r=4, cout=None):
    """per pixel L*(K+1)*4 read + 4 (coord) + L*K*4 written = 308 B at L=4, r=4.
    With `cout` (the lookup fused with the 1x1 layer that consumes it, dkt_corr1d_lookup_conv1x1): the
    L*K-channel lookup is never written; cout*4 B of the layer's output are: 160 + 4 + 256 = 420 B at cout=64."""
    K = 2 * r + 1
    return n_pixels * (L * (K + 1) * 4 + 4 + (cout if cout is not None else L * K) * 4)


class TimedCorr:
    """Wraps the corr object so every lookup launch is bracketed by HIP events on
    the stream it is launched on (torch's current stream == the stream handed to
    the C ABI)."""

    def __init__(self, fn, sink):
        self.fn, self.sink = fn, sink

    def __call__(self, coords):
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        out = self.fn(coords)
        b.record()
        self.sink.append((a, b))
        return out


def pmc_traffic(args):
    """HBM-side bytes per launch of the two reported kernels, measured now: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE
    in SEPARATE sub-runs (kernel trace only) of tools/pmc/pmc_probe.py, which runs this workload's forward with the loop's
    units as plain launches plus a known-traffic calibration stream; reads are FETCH_SIZE x the correction that stream yields."""
    import subprocess
    script = os.path.join(ROOT, "tools", "pmc", "run_pmc.sh")
    try:
        subprocess.run(["bash", script, str(args.batch)], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=2400)
        sfx = "" if args.batch == 1 else "_b%d" % args.batch
        with open(os.path.join(ROOT, "gpurun_out", "r06_pmc" + sfx, "traffic.json")) as f:
            j = json.load(f)
        return {"conv_bytes": j.get("gru_bytes"), "lookup_bytes": j.get("motion_front_bytes"),
                "lookup_operator_bytes": j.get("lookup_operator_bytes"), "note": "measured in this run: %s" % j.get("source")}
    except Exception as e:          # counters unavailable on this box: say so, report nothing
        return {"note": "--pmc failed: %s" % (str(e)[:200],)}


def conv_precision(dev):
    """Relative error of the path's convolution arithmetic against an fp64 convolution of the same operands, measured now
    (3x3, 128 -> 128, 96x160, standard-normal operands) -- what "22-bit operands" buys on this device."""
    import torch.nn.functional as F
    from dkt_stereo_amd import conv as _conv
    torch.manual_seed(5)
    layer = torch.nn.Conv2d(128, 128, 3, padding=1).to(dev)
    x = torch.randn(1, 128, 96, 160, device=dev)
    with torch.no_grad():
        want = F.conv2d(x.double(), layer.weight.double(), layer.bias.double(), padding=1)
        got = {"round-2 kernel (encoders)": _conv.conv2d(x, layer)}
        if _conv.get_backend() == "f16x3":
            from dkt_stereo_amd import conv_c8
            got["C8S kernel (refinement loop)"] = conv_c8.conv2d_c8([conv_c8.pack(x)], layer)
        return {k: float((v.double() - want).abs().max() / want.abs().max()) for k, v in got.items()}


def cpu_baseline(args, sd, cfg, i1, i2):
    """Reference-equivalent CPU path (oracle/torch_oracle.py) on this host's cores.
    Bounded sample: encoders + correlation build once, `cpu_iters` of the 32 GRU
    iterations, extrapolated to 32 (every iteration does identical work)."""
    from oracle import torch_oracle as to
    cores = torch.get_num_threads()
    sd_cpu = {k: v.cpu() for k, v in sd.items()}
    a, b = i1[:1].cpu(), i2[:1].cpu()
    with torch.no_grad():
        t0 = time.perf_counter()
        fmap1, fmap2, net, inp = to.raft_prepare(sd_cpu, cfg, a, b)
        t1 = time.perf_counter()
        to.raft_iterations(sd_cpu, cfg, fmap1, fmap2, net, inp, 1)          # warm-up (allocations, mkldnn primitives)
        t2 = time.perf_counter()
        to.raft_iterations(sd_cpu, cfg, fmap1, fmap2, net, inp, args.cpu_iters)
        t3 = time.perf_counter()
        to.raft_iterations(sd_cpu, cfg, fmap1, fmap2, net, inp, 1)
        t4 = time.perf_counter()
    once = t4 - t3                                  # corr build + 1 iteration + upsample
    per_iter = max((t3 - t2) - once, 1e-9) / max(args.cpu_iters - 1, 1)
    pair = (t1 - t0) + once + (args.iters - 1) * per_iter
    return {"value": 1.0 / pair, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": "1 pair %dx%d: encoders + corr build once, %d of %d GRU iterations timed "
                      "(%.2f s/iter), extrapolated to %d" % (args.height, args.width, args.cpu_iters,
                                                             args.iters, per_iter, args.iters),
            "s_per_pair": pair, "s_per_iter": per_iter}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0 and world != 1:
            print("warning: WORLD_SIZE %d != --gpus %d" % (world, args.gpus), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU path in the product")
    if args.dist_backend == "gloo":
        local_rank %= torch.cuda.device_count()          # (ranks may share a device; RCCL needs one device per rank)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import _synth
    from dkt_stereo_amd import _ffi
    from dkt_stereo_amd.raft_stereo import BASE_CONFIG, RAFTStereo
    from dkt_stereo_amd.shard import gather_disparity
    _ffi.lib()                                   # fail loudly if the HIP library is missing
    torch.backends.cudnn.benchmark = True       # MIOpen find mode, as tools/evaluate_stereo.py:113

    from dkt_stereo_amd import conv as _conv
    if args.conv_backend:
        _conv.set_backend(args.conv_backend)
    conv_backend_name = {"f16x3": "hip split-fp16 MFMA x3 (fp32-class, dkt_conv2d_f16s)",
                         "f16x2": "hip split-fp16 MFMA x2", "f16": "hip fp16 MFMA",
                         "miopen": "miopen-fp32"}[_conv.get_backend()]
    from dkt_stereo_amd.raft_stereo import make_args
    model = RAFTStereo(make_args(mixed_precision=True)) if args.mixed_precision else RAFTStereo()
    schedule = None
    if args.mixed_precision and not args.schedule:
        args.schedule = "%d,0" % args.iters          # (what mixed_precision runs: reported in `dtype` like any schedule)
    if args.schedule:
        k = [int(x) for x in args.schedule.split(",")]
        schedule = (k[0], k[1] if len(k) > 1 else 0)
        if sum(schedule) > args.iters or min(schedule) < 0:
            raise SystemExit("--schedule: K1 + K2 must not exceed --iters")
    model.precision_schedule = schedule if (schedule and sum(schedule)) else None
    sd = _synth.torch_state_dict(_synth.shapes_of(model), 7)
    model.load_state_dict(sd, strict=True)
    model.to(dev).eval()
    B = args.batch
    # every rank gets its own pairs (seeded by global pair index)
    pairs = [_synth.image_pair(1000 + rank * B + j, 1, args.height, args.width, 12 if j % 2 == 0 else 40)
             for j in range(B)]
    i1 = torch.cat([torch.from_numpy(p[0]) for p in pairs]).to(dev)
    i2 = torch.cat([torch.from_numpy(p[1]) for p in pairs]).to(dev)
    h4, w4 = args.height // 4, args.width // 4

    look_events = []
    import dkt_stereo_amd.raft_stereo as rs
    real_impls = dict(rs.CORR_IMPLEMENTATIONS)

    def timed_factory(cls):
        def make(*a, **k):
            return TimedCorr(cls(*a, **k), look_events)
        return make

    def step():
        _, up = model(i1, i2, iters=args.iters, test_mode=True)
        if world > 1:
            up = gather_disparity(up, B * world, dst=0)
        return up

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        # one-time state of the model for this shape, outside the warm-up and the timed region: the first forward packs
        # weights, picks the C8S scales and captures the loop; the second captures the encoder pass WITH the correlation
        # build (its persistent block exists from the first); the third lets the loop adopt the captured encoder pass's
        # outputs and captures its units once more.  From the fourth on every forward is the same replays.
        for _ in range(3):
            step()
        for _ in range(args.warmup):
            step()
        sync()
        model.check_finite = False          # no host sync per forward inside the timed region: checked once behind it
        t0 = time.perf_counter()
        for _ in range(args.steps):
            last = step()
        sync()
        t1 = time.perf_counter()
        model.check_finite = True
        gathered_batch = int(last.shape[0]) if last is not None else None      # (rank 0 holds all ranks' maps)
        # the timed region ran without the per-forward check (no host synchronisation inside it): its post-conditions are read
        # ONCE behind it -- the error word of the fused ConvGRU / chain launches (a time-out inside the region must not be
        # reported as a valid run), finiteness, the C8S scale window
        lp_t = (model._graph_state or {}).get("c8")
        error_word, ranges_ok = 0, True
        if lp_t is not None:
            s_t = lp_t.status(last if world == 1 else None)
            error_word, ranges_ok, finite_t = s_t.err, s_t.ranges_ok, s_t.finite
        else:
            finite_t = True
        if last is not None and not (finite_t and bool(torch.isfinite(last).all())):
            raise SystemExit("bench.py: non-finite disparities in the timed region")
        if error_word:
            raise SystemExit("bench.py: a flag-synchronised launch timed out inside the timed region (error word %d): the run is invalid"
                             % error_word)
        if not ranges_ok:
            raise SystemExit("bench.py: activations left the C8S scale window inside the timed region: the run is invalid")

        mine = t1 - t0
        # the same steps once more in the product's DEFAULT mode (check_finite = True: one status launch + one host
        # synchronisation per forward, repeats on a time-out / a left scale window): `value_default_mode`
        sync()
        td0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        mine_default = time.perf_counter() - td0
        # (and the unchecked steps once more behind it: the two modes are timed back to back on a device whose clocks drift with
        # its temperature -- `value_unchecked_again` shows how much of the difference is the order of the regions)
        model.check_finite = False
        tr0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        mine_again = time.perf_counter() - tr0
        model.check_finite = True
        # (gloo carries host tensors; RCCL device tensors)
        elapsed = torch.tensor([mine, mine_default], device=dev if args.dist_backend == "nccl" else "cpu", dtype=torch.float64)
        per_rank = [mine]
        if world > 1:
            every = [torch.zeros_like(elapsed) for _ in range(world)]
            dist.all_gather(every, elapsed)
            per_rank = [float(t[0].item()) for t in every]
            dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        elapsed, elapsed_default = float(elapsed[0].item()), float(elapsed[1].item())
        ranks_seen = dist.get_world_size() if world > 1 else 1
        backend = dist.get_backend() if world > 1 else "none (single process)"
        if rank != 0:
            # the instrumented passes below are rank 0's (no collective inside them)
            dist.destroy_process_group()
            return

        # N different pairs through the same model in default mode: a recalibration (new C8S scales +
        # re-capture + the pair again) can only happen when the input changes, which 20 steps on one pair never show
        distinct = None
        if args.distinct_pairs > 0 and world == 1:
            lp_d = (model._graph_state or {}).get("c8")
            before = (lp_d.recalibrations, lp_d.calibrations) if lp_d is not None else (0, 0)
            n_d = args.distinct_pairs
            times, amps = [], []
            for k in range(n_d):
                amp = 0.5 * 4.0 ** (k / max(n_d - 1, 1))              # x0.5 ... x2, geometric
                p1, p2 = _synth.image_pair(2000 + k, B, args.height, args.width, 12 if k % 2 == 0 else 40)
                a1, a2 = torch.from_numpy(p1).to(dev) * amp, torch.from_numpy(p2).to(dev) * amp
                torch.cuda.synchronize()
                tk = time.perf_counter()
                _, up_k = model(a1, a2, iters=args.iters, test_mode=True)
                torch.cuda.synchronize()
                times.append(time.perf_counter() - tk)
                amps.append(amp)
            lp_d = (model._graph_state or {}).get("c8")
            after = (lp_d.recalibrations, lp_d.calibrations) if lp_d is not None else (0, 0)
            ts = sorted(times)
            distinct = {"pairs": n_d, "amplitudes": [round(a, 3) for a in amps], "shifts": "12 / 40 alternating",
                        "value": B * n_d / sum(times), "unit": "pairs/s", "median_step_ms": 1e3 * ts[len(ts) // 2],
                        "worst_step_ms": 1e3 * ts[-1], "step_ms": [round(1e3 * t, 2) for t in times],
                        "recalibrations": after[0] - before[0],
                        "note": "default mode (check_finite = True), every step synchronised and timed on its own; a recalibration = "
                                "new C8S scales from the maxima the pair left behind + re-capture of the loop's units + the pair again"}
            model(i1, i2, iters=args.iters, test_mode=True)           # (back on the benchmark pair for the instrumented passes)
            torch.cuda.synchronize()

        # Per-kernel timing.  The timed steps above replay the GRU iteration from a captured
        # HIP graph, where a single kernel cannot be bracketed by events; the same workload is
        # therefore run once more through the eager path with an event pair around every
        # launch of (a) the dominant kernel -- the update-block convolution of the finest GRU,
        # gru08 z|r, 384->256 3x3 -- and (b) the correlation lookup (same stream, same inputs,
        # same kernels).  The cost of an empty event pair is measured and subtracted.
        c8_loop = model._graph_state is not None and model._graph_state.get("c8") is not None
        model.use_hip_graph = c8_loop        # (the C8S loop runs its units eagerly through model.c8_eager instead)
        look_events.clear()
        conv_events = []
        import dkt_stereo_amd.corr as dcorr
        import dkt_stereo_amd.update as upd
        real_gate_zr = upd.conv2d_gate_zr
        real_fused = dcorr.CorrBlock1D.lookup_conv1x1
        fused_lookup = model.fuse_lookup

        def timed_fused(self_, coords, layer, relu=True, tap=False, **kw):
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ea.record()
            y = real_fused(self_, coords, layer, relu=relu, tap=tap, **kw)
            eb.record()
            if y is not None:
                look_events.append((ea, eb))
            return y

        if fused_lookup:
            dcorr.CorrBlock1D.lookup_conv1x1 = timed_fused
        else:
            rs.CORR_IMPLEMENTATIONS = {k: timed_factory(v) for k, v in real_impls.items()}

        def timed_gate_zr(x, layer, cz, cr, h):
            # gru08: merged z|r convolution (384 -> 256, 3x3) with the gate epilogue, finest scale
            big = layer.weight.shape[0] == 256 and h.shape[2] == h4 and sum(t.shape[1] for t in x) == 384
            if not big:
                return real_gate_zr(x, layer, cz, cr, h)
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ea.record()
            y = real_gate_zr(x, layer, cz, cr, h)
            eb.record()
            conv_events.append((ea, eb))
            return y

        # round-3 loop (loop_c8.py): the same kernel is the paired launch gru08 z|r + gru32 z|r of conv_c8_kernel; its
        # units run eagerly here (model.c8_eager) with an event pair around that launch
        from dkt_stereo_amd import conv_c8 as dc8
        real_pair = dc8.launch_pair
        c8_used = []

        def timed_pair(d0, d1, ref, cfg):
            big = d0.epilogue == 1 and d0.Cout == 256 and d0.H == h4
            if not big:
                return real_pair(d0, d1, ref, cfg)
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ea.record()
            real_pair(d0, d1, ref, cfg)
            eb.record()
            conv_events.append((ea, eb))
            c8_used.append((d1.H, d1.W, d1.Cout, sum(d1.src_channels[i] for i in range(d1.nsrc)), cfg))

        # round-4 loop: the whole ConvGRU step (z|r -> gates -> q -> h') of gru08, with gru32 of the next iteration riding
        # along, is ONE launch of gru_c8_kernel (dkt_gru_c8_pair)
        real_gru = dc8.gru_launch
        gru_used = []

        def timed_gru(d0, d1=None, err=None, ref=None):
            if d0.H != h4 or d1 is None:
                return real_gru(d0, d1, err=err, ref=ref)
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ea.record()
            ok = real_gru(d0, d1, err=err, ref=ref)
            eb.record()
            if ok:
                conv_events.append((ea, eb))
                gru_used.append((sum(d0.x_channels[i] for i in range(d0.nx)), d1.H, d1.W, sum(d1.x_channels[i] for i in range(d1.nx))))
            return ok

        # the launch that carries the lookup inside the loop (dkt_motion_front_c8), timed where the loop runs it: the eager
        # units keep both streams busy exactly as the captured ones do, and the host runs ahead of the device, so an event
        # pair on the launch stream brackets the kernel's execution beside the middle GRU's chain
        real_front = dc8.motion_front
        front_events = []

        def timed_front(*a, **k):
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ea.record()
            real_front(*a, **k)
            eb.record()
            front_events.append((ea, eb))

        dc8.motion_front = timed_front
        dc8.gru_launch = timed_gru
        dc8.launch_pair = timed_pair
        upd.conv2d_gate_zr = timed_gate_zr
        model.c8_eager = c8_loop
        model(i1, i2, iters=args.iters, test_mode=True)
        torch.cuda.synchronize()
        model.c8_eager = False
        dc8.launch_pair = real_pair
        dc8.gru_launch = real_gru
        dc8.motion_front = real_front
        upd.conv2d_gate_zr = real_gate_zr
        dcorr.CorrBlock1D.lookup_conv1x1 = real_fused
        rs.CORR_IMPLEMENTATIONS = real_impls
        model.use_hip_graph = True
        empty = []
        for _ in range(64):
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ea.record()
            eb.record()
            empty.append((ea, eb))
        torch.cuda.synchronize()
        ev_overhead_ms = sorted(x.elapsed_time(y) for x, y in empty)[len(empty) // 2]
        look_ms = [max(a.elapsed_time(b) - ev_overhead_ms, 1e-6) for a, b in look_events]
        # The lookup kernel is shorter than the host cost of launching it from Python (15-20 us), so an event
        # pair around an eager launch mostly times the host.  It is therefore timed as the timed region runs it:
        # back-to-back launches replayed from a HIP graph on the launch stream (32 launches per replay, 8
        # replays), on the pyramid and coordinates the timed steps left behind.
        def graph_time_ms(one, n=32, reps=8):
            """Average duration of `one()`'s launch: n back-to-back launches per captured graph, `reps` replays."""
            for _ in range(3):
                one()
            torch.cuda.synchronize()
            gl = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gl):
                for _ in range(n):
                    one()
            gl.replay()
            torch.cuda.synchronize()
            ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ea.record()
            for _ in range(reps):
                gl.replay()
            eb.record()
            torch.cuda.synchronize()
            return ea.elapsed_time(eb) / (n * reps)

        st = model._graph_state
        front_ms = op_ms = None
        if st is not None:
            blk_, c1_ = st["corr"], st["coords1"]
            layer_ = model.update_block.encoder.convc1
            if c8_loop:                    # as the loop calls it: straight into the C8S operand of convc2
                cor_ = st["c8"].cor
                one = lambda: blk_.lookup_conv1x1(c1_, layer_, out_c8=cor_)   # noqa: E731
            elif fused_lookup:
                one = lambda: blk_.lookup_conv1x1(c1_, layer_)          # noqa: E731
            else:
                one = lambda: blk_(c1_)                                 # noqa: E731
            look_ms = [graph_time_ms(one)] * 256
            # the reference-visible operator corr_fn(coords) (core/corr.py:127-146) on the same pyramid and coordinates
            op_ms = graph_time_ms(lambda: blk_(c1_))
            lp0 = st.get("c8")
            if lp0 is not None and getattr(lp0, "front", False):
                # the launch the loop makes (dkt_motion_front_c8), stand-alone: the head's planes of the last iteration, the
                # coordinate buffers alternating as in the loop (two launches per call: the state returns to its parity)
                fh_, enc_ = model.update_block.flow_head, model.update_block.encoder
                from dkt_stereo_amd.update import _leading_outputs
                planes_, nco_ = dc8.head_planes([lp0.hc8[0]], fh_.conv1, _leading_outputs(fh_.conv2, 1), cfg=lp0.cfg["head"])
                planes_.zero_()            # (delta = bias only: the coordinate stays in range over the 512 timed launches)
                cx_ = lp0._coords(st)
                keep_ = [t.clone() for t in (cx_[0], cx_[1], st["flow"])]

                def two_fronts():
                    for p_ in (0, 1):
                        real_front(st["corr"], planes_, nco_, None, cx_[p_], cx_[1 - p_], st["coords0"][:, :1], st["flow"],
                                   enc_.convc1, lp0.cor, enc_.convf1, lp0.flo)

                front_ms = graph_time_ms(two_fronts, n=16) / 2.0
                for t, k in zip((cx_[0], cx_[1], st["flow"]), keep_):
                    t.copy_(k)
        front_pipe_ms = [max(a.elapsed_time(b) - ev_overhead_ms, 1e-6) for a, b in front_events]
        conv_ms = [max(a.elapsed_time(b) - ev_overhead_ms, 1e-6) for a, b in conv_events]

        # hot path alone (what the C ABI covers + the update block), encoders excluded
        fmap1, fmap2, net, inp = model.encode(i1, i2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        model.iterate(fmap1, fmap2, net, inp, args.iters)
        e1.record()
        torch.cuda.synchronize()
        hot_ms = e0.elapsed_time(e1)

    lp_ = (model._graph_state or {}).get("c8")
    front_used = bool(lp_ is not None and getattr(lp_, "front", False))
    precision = {"operand_bits": {"f16x3": 22, "f16x2": 11, "f16": 11, "miopen": 24}[_conv.get_backend()],
                 "accumulate": "fp32",
                 "conv_rel_err_vs_fp64": conv_precision(dev),
                 "vendor_fp32_conv_rel_err_vs_fp64": 0.9e-6,
                 "bound": "final disparity <= 1e-3 max-abs of the reference (north_star); see max_abs_vs_reference"}
    n_pix = B * h4 * w4
    look_avg_ms = sum(look_ms) / max(len(look_ms), 1)
    alg = lookup_bytes_per_launch(n_pix, cout=64 if fused_lookup else None)
    achieved = alg / (look_avg_ms * 1e-3) / 1e9 if look_avg_ms > 0 else 0.0
    passes = {"f16x3": 3, "f16x2": 2, "f16": 1}.get(_conv.get_backend(), 1)
    if model.precision_schedule:        # (average MFMA products per block over the timed launches of a scheduled loop)
        passes = (schedule[0] + 2 * schedule[1] + 3 * (args.iters - sum(schedule))) / float(args.iters)
    conv_avg_ms = sum(conv_ms) / max(len(conv_ms), 1)
    conv_alg_flops = 2.0 * n_pix * 384 * 9 * 256
    conv_kernel = "conv2d_f16s_kernel (dkt_conv2d_f16s_gate_zr), gru08 z|r 384->256 3x3 + gate epilogue @%dx%d" % (h4, w4)
    if c8_used:
        # the paired launch also carries the coarsest GRU's z|r convolution (its tiles ride in the same launch)
        h2, w2, co2, ci2, cfg2 = c8_used[0]
        conv_alg_flops += 2.0 * B * h2 * w2 * ci2 * 9 * co2
        conv_kernel = ("conv_c8_kernel<4,2,4,4> (dkt_conv2d_c8_pair, tile shape %d): gru08 z|r 384->256 3x3 @%dx%d + gru32 z|r "
                       "%d->%d @%dx%d, gate epilogues, C8S operands" % (cfg2, h4, w4, ci2, co2, h2, w2))
    if gru_used:
        # one launch = both convolutions of the finest ConvGRU (z|r: 128 + x -> 256, q: 128 + x -> 128) and of the coarsest
        xc, h2, w2, xc2 = gru_used[0]
        conv_alg_flops = 2.0 * n_pix * (128 + xc) * 9 * 384 + 2.0 * B * h2 * w2 * (128 + xc2) * 9 * 384
        conv_kernel = ("gru_c8_kernel (dkt_gru_c8_pair): one ConvGRU step per launch -- gru08 z|r %d->256 + gates + q %d->128 + "
                       "state update @%dx%d, with gru32 (%d->256, %d->128 @%dx%d) riding along; C8S operands, z in registers, "
                       "neighbour-tile flags instead of a kernel boundary" % (128 + xc, 128 + xc, h4, w4, 128 + xc2, 128 + xc2, h2, w2))
    conv_tflops_exec = passes * conv_alg_flops / (conv_avg_ms * 1e-3) / 1e12 if conv_avg_ms > 0 else 0.0
    # HBM-side bytes per launch from the committed PMC passes (FETCH_SIZE + WRITE_SIZE, separate
    # rocprofv3 runs of the same kernels on the same shapes); None when the file is absent
    # The committed passes are quoted ONLY when they profiled the launch timed here: same image size and batch, fused ConvGRU
    # launch with the same rider (profiles/r06_hbm_traffic[_b<B>].json record both)
    traffic = {}
    try:
        with open(os.path.join(ROOT, "profiles", "r06_hbm_traffic%s.json" % ("" if B == 1 else "_b%d" % B))) as f:
            traffic = json.load(f)
    except (OSError, ValueError):
        pass
    default_shape = (args.height, args.width) == (736, 1248) and world == 1
    traffic_applies = bool(traffic and gru_used and default_shape and traffic.get("shape") == [args.height, args.width, B]
                           and traffic.get("gru_rider_hw") == [gru_used[0][1], gru_used[0][2]])
    live_traffic = pmc_traffic(args) if (args.pmc and default_shape and gru_used) else {}
    out = {
        "metric": "stereo pairs/sec at 736x1248 D=192, 32 iters (RAFT-Stereo test_mode forward)",
        "value": world * B * args.steps / elapsed,
        "unit": "pairs/s",
        "n_gpus": world,
        "ranks_seen": ranks_seen,
        "gathered_batch": gathered_batch,
        "collective_backend": backend,
        "per_rank_pairs_per_s": [B * args.steps / t for t in per_rank],
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        # the timed region runs with check_finite = False (no host synchronisation inside it; its post-conditions -- error word,
        # finiteness, scale window -- are read once behind it: `error_word`); the same steps in the product's default mode:
        "value_default_mode": world * B * args.steps / elapsed_default,
        "ms_per_step_default_mode": 1e3 * elapsed_default / args.steps,
        "value_unchecked_again": world * B * args.steps / mine_again,
        "error_word": error_word,
        "recalibrations": (distinct or {}).get("recalibrations"),
        "distinct_pairs": distinct,
        "ms_per_iter": hot_ms / args.iters,
        "hot_path_ms_per_pair": hot_ms / B,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        # the arithmetic type of the path, not a precision claim: tensors and accumulation are fp32, every product of the
        # convolutions is evaluated on the fp16 matrix pipe from split operands (see "precision")
        "dtype": ("f32 io/accumulate; REDUCED-PRECISION SCHEDULE (not the headline): of the %d refinement iterations the first %d with weights "
                  "and activations rounded to fp16 (1 MFMA product), the next %d with activations rounded to fp16 (2 products), the last "
                  "%d fp32-class (3 products); %s"
                  % (args.iters, schedule[0], schedule[1], args.iters - sum(schedule),
                     "args.mixed_precision = True (raft_stereo.py:95,156): the encoders' convolutions at one fp16 product as well; correlation "
                     "volume, lookup, up-sampling exact fp32" if args.mixed_precision else
                     "encoders, correlation volume, lookup, up-sampling fp32-class / exact fp32"))
                 if model.precision_schedule else
                 {"f16x3": "f32 io/accumulate; convolution products = 3x fp16-split MFMA (22-bit operands); lookup / correlation exact fp32",
                  "f16x2": "f32 io/accumulate; convolution products = 2x fp16-split MFMA (NOT a parity path)",
                  "f16": "f32 io/accumulate; convolution products = fp16 MFMA (NOT a parity path)",
                  "miopen": "f32 (vendor fp32 convolutions)"}[_conv.get_backend()],
        "precision": precision,
        "precision_schedule": ({"one_pass_iterations": schedule[0], "two_pass_iterations": schedule[1],
                                "note": "errors against the reference fixtures per schedule: profiles/r05_precision_schedule.txt"}
                               if model.precision_schedule else None),
        "data": "synthetic (seeded U[0,255) left image, shifted+noised right image; random-init weights)",
        "config": {"workload": "RAFT-Stereo %dx%d (1/4 res %dx%d), D=192, %d GRU iters, batch %d/GPU, "
                               "corr_implementation=reg, BASELINE.json configs[1]"
                               % (args.height, args.width, h4, w4, args.iters, B),
                   "parallelism": "dp%d (independent pairs per rank, result gather only)" % world,
                   # the N = 1 line is BASELINE.json configs[1] (batch 1); N > 1 runs configs[3]'s per-GPU share (batch 8 per
                   # GPU): its weak-scaling reference is `--gpus 1 --batch 8` (profiles/r03_bench_lines.jsonl, line 2)
                   "per_gpu_batch": B,
                   "weak_scaling_reference": "bench.py --gpus 1 --batch %d" % B,
                   "conv_backend": (conv_backend_name.replace("dkt_conv2d_f16s", "refinement loop: dkt_conv2d_c8, encoders: dkt_conv2d_f16s incl. the weights-stationary 64->64 kernel")
                                    if (c8_used or gru_used) else conv_backend_name),
                   "loop": ("C8S convolutions (loop_c8.py), fused ConvGRU launch" if gru_used else
                            "C8S convolutions (loop_c8.py)" if c8_used else "round-2 kernels")},
        # dominant kernel (~70 % of a pair): the split-fp16 implicit-GEMM convolution.  It is
        # MFMA-bound; `achieved` counts the fp16 MFMA flops it executes per launch
        # (passes x 2*H*W*Cin*9*Cout; the fp32-equivalent algorithmic figure is 1/passes of it)
        # against the dense fp16 MFMA peak.
        # `achieved` / `frac` are ALGORITHMIC: the layer's 2*H*W*Cin*9*Cout flops (what an fp32 convolution
        # of this shape needs) per launch over the launch time, against the dense fp16 MFMA peak the kernel
        # issues on.  The kernel spends `mfma_passes` fp16 MFMAs per algorithmic product (split operands,
        # fp32-class result): `mfma_issue_frac` = passes x frac is how busy it keeps the matrix pipe.
        "roofline": {"kernel": conv_kernel,
                     "bound": "mfma", "achieved": conv_tflops_exec / passes, "peak": FP16_MFMA_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": conv_tflops_exec / passes / FP16_MFMA_PEAK_TFLOPS,
                     "frac_algorithmic": conv_tflops_exec / passes / FP16_MFMA_PEAK_TFLOPS,
                     "mfma_issue_frac": conv_tflops_exec / FP16_MFMA_PEAK_TFLOPS,
                     "mfma_issued_tflops": conv_tflops_exec,
                     # HBM-side bytes per launch: measured in THIS run only with --pmc (two rocprofv3 --pmc sub-runs of the same
                     # launch); otherwise null, and the builder's committed passes are quoted under their own key
                     "traffic": live_traffic.get("conv_bytes"),
                     "traffic_note": live_traffic.get("note", "not measured in this run (pass --pmc)"),
                     "traffic_from_profiles": ({"bytes": traffic.get("gru_bytes"), "compulsory_bytes": traffic.get("gru_compulsory_bytes"),
                                                "source": "%s -- NOT this run" % traffic.get("source")}
                                               if traffic_applies else None),
                     "algorithmic_flops_per_launch": conv_alg_flops, "mfma_passes": passes,
                     "avg_launch_us": 1e3 * conv_avg_ms, "launches_timed": len(conv_ms),
                     "event_pair_overhead_us": 1e3 * ev_overhead_ms},
        # the kernel BASELINE.json's north_star sets the HBM target for: see lookup_rooflines() below
        "roofline_lookup": None,
    }
    # ---- the corr-lookup kernel (HBM-bound; north_star: >= 60 % of HBM), three forms, all timed live in this run:
    #   (1) what the loop launches: motion_front_kernel (coordinate update + lookup + convc1 + 7x7 stem), stand-alone
    #       (graph-replayed back to back) AND in the pipeline (event pairs around the loop's own launches, beside the middle
    #       GRU's chain on the other stream);
    #   (2) the lookup fused with convc1 alone (corr_feat64_kernel), the round-3 form the front embeds;
    #   (3) the reference-visible operator corr_fn(coords) (corr1d_lookup_skew_kernel).
    def hbm(alg_bytes, ms):
        ach = alg_bytes / (ms * 1e-3) / 1e9 if ms and ms > 0 else 0.0
        return {"achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_us": 1e3 * ms if ms else None}

    alg_op = lookup_bytes_per_launch(n_pix)
    # front: the fused lookup's 420 B/pixel + the head's 18 planes (72) + old coordinate, reference coordinate (8) read, new
    # coordinate and flow (8) written + the stem's 64 C8S channels (256) written = 764 B/pixel
    alg_front = n_pix * (420 + 72 + 8 + 8 + 256)
    fused_entry = dict(hbm(alg, look_avg_ms), kernel=("corr_feat64_kernel<4> (dkt_corr1d_lookup_conv1x1%s): pyramid lookup fused with "
                       "the motion encoder's 1x1 layer, 36 -> 64 channels, exact-fp32 MFMA" % ("_c8, C8S output" if (c8_used or gru_used) else ""))
                       if fused_lookup else "corr1d_lookup_skew_kernel<4> (dkt_corr1d_lookup_skew)", launches_timed=len(look_ms))
    op_entry = (dict(hbm(alg_op, op_ms), kernel="corr1d_lookup_skew_kernel<4> (dkt_corr1d_lookup_skew): the reference-visible "
                     "CorrBlock1D.__call__ (core/corr.py:127-146), 308 B/pixel", launches_timed=256) if op_ms else None)
    if front_used and front_ms:
        pipe_ms = sum(front_pipe_ms) / max(len(front_pipe_ms), 1) if front_pipe_ms else None
        rl = dict(hbm(alg_front, front_ms), bound="hbm",
                  kernel="motion_front_kernel<4> (dkt_motion_front_c8): what the loop launches once per iteration -- coordinate update "
                         "behind the flow head + pyramid lookup + convc1 (36 -> 64, exact-fp32 MFMA) + the motion encoder's 7x7 stem, "
                         "C8S outputs; 764 B/pixel algorithmic", launches_timed=256,
                  in_pipeline_avg_launch_us=1e3 * pipe_ms if pipe_ms else None,
                  in_pipeline_frac=(alg_front / (pipe_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if pipe_ms else None,
                  in_pipeline_launches_timed=len(front_pipe_ms),
                  in_pipeline_note="event pairs around the loop's own launches in this run (eager units, both streams busy as in the "
                                   "captured loop); the launch shares the device with the middle GRU's z|r convolution",
                  fused_lookup_conv1x1=fused_entry, reference_operator=op_entry)
    else:
        rl = dict(fused_entry, bound="hbm", in_loop="its own launch", reference_operator=op_entry)
    rl["traffic"] = live_traffic.get("lookup_bytes")
    rl["traffic_note"] = live_traffic.get("note", "not measured in this run (pass --pmc)")
    rl["traffic_kernel"] = "motion_front_kernel (the loop's launch)"
    rl["traffic_from_profiles"] = ({"bytes": traffic.get("motion_front_bytes"), "reference_operator_bytes": traffic.get("lookup_operator_bytes"),
                                    "source": "%s -- NOT this run" % traffic.get("source")}
                                   if traffic_applies and front_used else None)
    out["roofline_lookup"] = rl
    # EPE against the reference itself (BASELINE.json's "EPE vs ref"): tests/golden/raft_e2e.npz holds the
    # reference's CPU output for this exact workload (seed 3, shift 40, 32 iterations, every 8th pixel)
    if default_shape and B == 1 and args.iters == 32:
        try:
            import numpy as np
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import _cases
            c = _cases.E2E_CASES["736x1248_it32"]
            g = np.load(os.path.join(ROOT, "tests", "golden", "raft_e2e.npz"))
            r1, r2 = _synth.image_pair(c["seed"], 1, c["H"], c["W"], c["shift"])
            with torch.no_grad():
                _, up = model(torch.from_numpy(r1).to(dev), torch.from_numpy(r2).to(dev), iters=32, test_mode=True)
            st = int(g["736x1248_it32/stride"])
            diff = np.abs(up[:, :, ::st, ::st].cpu().numpy() - g["736x1248_it32/flow_up"])
            out["epe_vs_reference"] = float(diff.mean())
            out["max_abs_vs_reference"] = float(diff.max())
            out["reference_fixture"] = "tests/golden/raft_e2e.npz:736x1248_it32 (jiaw-z/DKT-Stereo RAFTStereo.forward on CPU, fp32)"
        except (OSError, KeyError, ImportError) as e:          # fixture absent: report nothing rather than fail
            out["epe_vs_reference"] = None
            out["reference_fixture"] = "unavailable: %s" % e
    if not args.skip_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(args, sd, dict(BASE_CONFIG), i1, i2)
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
