"""The RAFT-Stereo refinement loop (meta_arch/raft_stereo/raft_stereo.py:146-167 with the update block of
core/update.py:97-138) on the round-3 convolution (csrc/conv_c8.hip): every 3x3 layer of the update block reads
pre-split "C8S" activations that its producers write from their epilogues, hidden states are kept twice (fp32 NCHW for the
gate arithmetic / resampling, C8S for the convolutions).

Schedule (the rotated one of RAFTStereo._iterate_rotated, on one stream -- kernels on different streams do not
overlap usefully on this part, tools/concurrency_probe.py):
    prologue: gru32(0)
    unit(i) : gru16(i) ; lookup + motion encoder(i) ; gru08(i) paired with gru32(i+1) ; flow head(i)
Every GRU sees exactly the operands of the reference's sequential order."""
import math
import os

import torch

from . import _ffi
from . import conv as _conv
from . import conv_c8 as c8
from .update import interp, pool2x, _leading_outputs, _scaled_layer

#: tile shapes (conv_c8.hip c8_dispatch) per layer class
_CFG = dict(zr08=1, q08=2, zr16=4, q16=4, head=2, enc=3, c2=4)
def _env_cfg():
    """DKT_C8_CFG="zr16=1,q16=2": tile shapes per layer class for A/B runs (conv_c8.hip c8_dispatch)."""
    out = {}
    for kv in os.environ.get("DKT_C8_CFG", "").split(","):
        if "=" in kv:
            k, v = kv.split("=")
            out[k.strip()] = int(v)
    return out


def _cfg_for_batch(B, n1):
    """Tile shapes that depend on how many tiles the batch gives (round 5).  _CFG's 4-row / 64-channel tiles for the middle GRU
    and the 64-channel pair exist to put enough blocks on the device at ONE pair per launch; with several pairs per launch the
    wide 8-row tiles (fewer fragment reads per MFMA) have enough tiles of their own."""
    out = {}
    tiles8 = B * ((n1.shape[2] + 7) // 8) * ((n1.shape[3] + 31) // 32)
    if tiles8 >= _WIDE_MID_TILES:
        out.update(_CFG_WIDE_MID)
    out.update(_env_cfg())
    return out


#: 8 x 32 tiles of the middle level (x batch) from which its GRU takes the wide tiles of _CFG_WIDE_MID (measured at cfg4's 8 pairs
#: per launch, profiles/r05_b8_variants.txt); DKT_C8_WIDE_MID_TILES overrides
_WIDE_MID_TILES = int(os.environ.get("DKT_C8_WIDE_MID_TILES", "400"))
_CFG_WIDE_MID = dict(zr16=1, q16=2)

#: flow head with the hidden tensor reduced in conv1's epilogue (FUSE_HEAD = False: hidden tensor + few-output kernel)
FUSE_HEAD = True


#: round 4: the finest GRU (with the coarsest one of the next iteration riding along) as ONE launch per step (csrc/gru_c8.hip:
#: z|r -> gates -> q -> h' per tile, neighbour flags instead of a kernel boundary); DKT_C8_FUSE_GRU=0: two launches per step
FUSE_GRU = os.environ.get("DKT_C8_FUSE_GRU", "1") != "0"
#: 8 x 32 tiles of the finest level (x batch) from which the one-launch form is taken; below: two launches on the tiles of _CFG_SMALL
FUSE_GRU_MIN_TILES = 128
_CFG_SMALL = dict(zr08=4, q08=4)

#: round 4: every C8S tensor of the loop carries a power-of-two scale picked from the magnitudes one trial unit produces
#: (C8Loop.calibrate): max |x * scale| lands in [2^SCALE_EXP, 2^(SCALE_EXP + 1)), i.e. 5 bits below fp16's overflow and with
#: hi AND lo halves in fp16's normal range down to 2^-13 of the tensor's maximum.  DKT_C8_AUTOSCALE=0: scale 1 everywhere
AUTOSCALE = os.environ.get("DKT_C8_AUTOSCALE", "1") != "0"
SCALE_EXP = 10
#: a later forward whose maxima leave [2^RANGE_LO, 2^RANGE_HI) after scaling recalibrates (checked where finiteness is)
RANGE_LO, RANGE_HI = 6, 14

#: the middle GRU's chain of the next iteration on a second stream beside the head and the motion encoder (DKT_C8_FORK=0: one stream)
#: (=2: also the motion encoder's 7x7 stem beside the lookup on a third stream -- measured 0.25 ms per pair SLOWER, kept for A/B)
FORK = int(os.environ.get("DKT_C8_FORK", "1"))
#: the prologue's motion features beside its hidden-state chain (see C8Loop.prologue)
PROLOGUE_FORK = True
#: the coordinate update behind the flow head, the lookup + convc1 and the motion encoder's 7x7 stem as ONE launch
#: (dkt_motion_front_c8; the x coordinate alternates between two buffers, see C8Loop.unit)
FRONT = True
#: the two resampling jobs in front of / behind the middle GRU as one launch each (dkt_resample_pair_c8)
PAIR_RESAMPLE = True
#: the flow head is the fused GRU launch's first successor in the captured unit (see C8Loop.unit)
HEAD_FIRST = True
#: round 5 prototype: dependent layer pairs of the window between two fused-GRU launches as ONE
#: launch each, a flag round instead of the kernel boundary (dkt_conv2d_c8_chain): convc2 | convf2 -> encoder.conv on the main
#: queue, the middle GRU's z|r -> q on the forked one.  0 = off (default: measured, profiles/r05_chain.txt), 1 = on,
#: 2 = timing only (no waits: the upper bound of what the fusion can buy; results are wrong).  Both chains can be in flight at
#: once, and every block of a chain must stay resident while it waits: CHAIN_BLOCKS each
CHAIN = int(os.environ.get("DKT_C8_CHAIN", "0"))
CHAIN_BLOCKS = int(os.environ.get("DKT_C8_CHAIN_BLOCKS", "256"))
#: units per captured graph: a replay boundary costs ~8 us of idle device between two units (profiles/r04_pair_breakdown.txt), so
#: the bulk of the iterations is replayed GRAPH_UNITS units at a time (a one-unit graph serves the remainder)
GRAPH_UNITS = max(1, int(os.environ.get("DKT_C8_GRAPH_UNITS", "8")))

#: round 5 -- precision schedule of the loop's convolutions (csrc/conv_c8.hip, gru_c8.hip `passes`): (k1, k2) = the first k1 units at
#: ONE fp16 MFMA product per block (weights and activations rounded to fp16), the next k2 at TWO (activations rounded), the rest at
#: the fp32-class three.  None = three throughout: the parity path and the default.  The refinement is contractive (SURVEY 7:
#: noise of 1e-4 on every lookup moves the result by 3.8e-5), so the arithmetic of the early iterations -- which only have to
#: bring the disparity close -- is largely forgotten by the final fp32-class ones; what a schedule costs against the reference
#: fixtures is measured in profiles/r05_precision_schedule.txt (tools/precision_schedule.py).  The reference's own switch of this
#: kind is `mixed_precision` (raft_stereo.py:95,156).  DKT_C8_SCHEDULE="k1,k2" sets the default.
def _env_schedule():
    v = os.environ.get("DKT_C8_SCHEDULE", "")
    if not v:
        return None
    k = [int(x) for x in v.split(",")]
    return (k[0], k[1] if len(k) > 1 else 0)


SCHEDULE = _env_schedule()

#: quarter-resolution pixels (per pair) from which the loop takes this path.  Round 4: every size does (small images on the
#: two-launch form of the finest GRU with 4-row tiles, see C8Loop.__init__): 256 x 512 / 32 iterations 9.5 ms against 12.1 on
#: the round-2 kernels; the variable remains as the A/B handle
MIN_PIXELS = int(os.environ.get("DKT_C8_MIN_PIXELS", "0"))


def eligible(model, shape=None):
    """RAFT-Stereo's update block (core/update.py:97-138) on this loop."""
    return _eligible_block(model.update_block, shape, 126)


#: IGEV would keep the round-2 loop below this many quarter-resolution pixels.  0 since the end of round 4: every size measured
#: takes this loop (tools/igev_small_shapes.py, profiles/r04_igev_small_shapes.txt: 64x128 ... 160x256 at 1/4 resolution
#: 8.6 ... 20.9 ms per 32 iterations against 11.5 ... 28.9 on the round-2 loop, results 1e-5 apart)
IGEV_MIN_PIXELS = 0


def eligible_igev(ub, shape=None):
    """IGEV's update block (meta_arch/igev_stereo/update.py:94-142: gru04 / gru08 / gru16, 127-channel motion encoder,
    one-output disparity head) on this loop."""
    return (_eligible_block(ub, shape, 127, max(MIN_PIXELS, IGEV_MIN_PIXELS))
            and all(hasattr(ub, n) for n in ("gru04", "gru08", "gru16", "disp_head")) and ub.disp_head.conv2.weight.shape[0] == 1)


def _eligible_block(ub, shape, enc_out, min_pixels=None):
    a = ub.args
    if shape is not None and shape[2] * shape[3] < (MIN_PIXELS if min_pixels is None else min_pixels):
        return False
    if a.n_gru_layers != 3 or getattr(a, "slow_fast_gru", False) or _conv.get_backend() != "f16x3":
        return False
    if any(getattr(m, "dkt_in_exp", 0) for m in ub.modules()):
        return False                      # calibrated activation exponents: the round-2 kernels handle those
    hd = list(a.hidden_dims)
    enc = ub.encoder
    stem = getattr(enc, "convf1", None) or getattr(enc, "convd1", None)
    # (the loop's C8S buffers are laid out for the reference's layer widths: 64 + 64 -> 128 motion features)
    return (hd == [128, 128, 128] and enc.conv.weight.shape[0] == enc_out and enc.convc1.weight.shape[0] == 64
            and stem is not None and stem.weight.shape[0] == 64 and tuple(stem.weight.shape[2:]) == (7, 7)
            and enc.convc2.weight.shape[:2] == (64, 64))


class LoopStatus:
    """What C8Loop.status read back: the error word, finiteness of the result, the scale window, the maxima behind it."""
    __slots__ = ("err", "finite", "ranges_ok", "maxima")

    def __init__(self, err, finite, ranges_ok, maxima):
        self.err, self.finite, self.ranges_ok, self.maxima = err, finite, ranges_ok, maxima


class C8Loop:
    """Buffers and the captured unit for one (shape, weights) state of RAFTStereo._iterate_graphed."""

    def __init__(self, model, st):
        self.ub = ub = getattr(model, "update_block", model)
        self.grus = self._gru_modules(ub)                # finest, middle, coarsest
        n0, n1, n2 = st["net"]
        dev = n0.device
        B = n0.shape[0]
        A = lambda t, C=128: c8.ActC8(B, C, t.shape[2], t.shape[3], dev)
        self.hc8 = [A(n0), A(n1), A(n2)]
        self.rh = [A(n0), A(n1), A(n2)]
        self.up1 = A(n0)                                  # gru08 operands: interp(net[1]), motion features (+ flow / disparity tail)
        self.mf = c8.ActC8(B, 128, n0.shape[2], n0.shape[3], dev, tail=self._tail_channels)
        self.pool0, self.up2 = A(n1), A(n1)               # gru16 operands: pool2x(net[0]), interp(net[2])
        self.pool1 = A(n2)                                # gru32 operand: pool2x(net[1])
        self.cor, self.flo = A(n0, 64), A(n0, 64)
        self.cf = A(n0, 128)
        self.hidden = torch.empty((B, 256, n0.shape[2], n0.shape[3]), device=dev, dtype=torch.float32)
        # fused ConvGRU launches: per-tile flag words per level, one error word (a neighbour wait that timed out)
        self.gflags = [c8.gru_flags(B, n.shape[2], n.shape[3], dev) for n in (n0, n1, n2)]
        self.err = torch.zeros(1, device=dev, dtype=torch.int32)
        # Tile shapes per layer class, and the form of the finest ConvGRU, by the number of 8 x 32 tiles the level gives: the
        # one-launch form holds ONE tile per CU through both convolutions, so an image of 32 tiles (256 x 512) keeps 32 CUs busy
        # for a whole step; below FUSE_GRU_MIN_TILES the step is two launches on 4-row tiles (9.5 ms against 15.0 at
        # 256 x 512 / 32 iterations, 14.8 against 17.5 at 480 x 640; 544 x 960 -- 136 tiles -- and up: the one launch wins)
        tiles = B * ((n0.shape[2] + 7) // 8) * ((n0.shape[3] + 31) // 32)
        self.cfg = dict(_CFG)
        self.cfg.update(_cfg_for_batch(B, n1))
        self.fuse_gru = FUSE_GRU and tiles >= FUSE_GRU_MIN_TILES
        if not self.fuse_gru and FUSE_GRU:
            self.cfg.update(zr08=_CFG_SMALL["zr08"], q08=_CFG_SMALL["q08"])
        # captured units, per parity of the coordinate buffer they start from (self.par; always 0 without the fused front)
        self.graph = None                # [parity] one unit
        self.graph_n = None              # [parity] GRAPH_UNITS units
        self.graph_last = None           # [parity] the final unit of a pair
        self.pinned = []                 # packed weight images the captured units point to (see capture)
        self.unit_launches = None
        self.chain_flags, self.chain_z = {}, {}          # dkt_conv2d_c8_chain: flag words per layer pair, z of the middle GRU
        self.chain_ok = True
        if CHAIN == 2:
            import warnings
            warnings.warn("DKT_C8_CHAIN=2: chain launches run WITHOUT their waits (a timing experiment): results are wrong")
        self.front = bool(FRONT) and self._front_supported(st)
        mask = getattr(ub, "mask", None)               # (IGEV's mask features are computed by its caller)
        self.mask_head = mask if (mask is not None and len(mask) == 3 and isinstance(mask[0], torch.nn.Conv2d)
                                  and tuple(mask[0].weight.shape[2:]) == (3, 3) and tuple(mask[2].weight.shape[2:]) == (1, 1)) else None
        self.mask_out = None
        self.par = 0                     # which of self.cx holds the current x coordinate
        self.cx = None
        self.calibrated = not AUTOSCALE
        self.schedule = SCHEDULE         # (k1, k2) or None, see SCHEDULE
        self.calibrations = 0            # trial runs (calibrate) and rescales without a trial (rescale_from) so far
        self.recalibrations = 0          # of those: forced by a checked forward whose maxima had left the window

    def plan(self, iters):
        """MFMA passes of each of the `iters` units."""
        if not self.schedule:
            return [3] * iters
        k1, k2 = self.schedule
        return [1 if i < k1 else 2 if i < k1 + k2 else 3 for i in range(iters)]

    # ---- captured units -------------------------------------------------------------------------------------------
    def _front_supported(self, st):
        from . import conv_c8
        return "corr" in st and conv_c8.motion_front_supported(st["corr"], self.ub.encoder)

    def _captured(self, kind, passes, par, st, capture_graph):
        """The captured graph of `kind` ("one" unit, the "last" unit of a pair, "n" = GRAPH_UNITS units back to back) at `passes`
        MFMA products per block, starting from coordinate-buffer parity `par` -- captured on first use (round 6: a pair touches
        four of the six (kind, parity) combinations, and a recalibration pays for every capture it triggers).  Capturing
        records, it does not execute."""
        if self.graph is None:
            self.graph, self.graph_last, self.graph_n = {}, {}, {}
            self.pinned = []             # the packed weight images the captured launches point to (conv_c8.pin_packs)
        store = {"one": self.graph, "last": self.graph_last, "n": self.graph_n}[kind]
        g = store.get((passes, par))
        if g is not None:
            return g
        if not getattr(self, "_mask_packed", None) == (passes, id(self.pinned)):
            with c8.passes(passes):
                self._mask(st)      # eager once: the final unit's own layers are packed (with the calibrated scales) outside the capture
            self._mask_packed = (passes, id(self.pinned))
        torch.cuda.synchronize()
        keep = self.par
        self.par = par
        g = torch.cuda.CUDAGraph()
        with _ffi.launch_log() as names:
            with capture_graph(g), c8.pin_packs(self.pinned), c8.passes(passes):
                if kind == "one":
                    self.unit(st)
                elif kind == "last":
                    self.unit(st, last=True)
                else:
                    for _ in range(GRAPH_UNITS):
                        self.unit(st)
        if kind == "one" or (kind == "n" and self.unit_launches is None):
            names = list(names)
            self.unit_launches = names if kind == "one" else names[:len(names) // GRAPH_UNITS]   # (tests pin their number)
        self.par = keep
        store[passes, par] = g
        return g

    def capture(self, st, capture_graph, passes=3):
        """Every kind of captured unit at `passes` products, both parities (what rounds 4-5 did up front; tools and tests)."""
        for p in ((0, 1) if self.front else (0,)):
            for kind in ("one", "last") + (("n",) if GRAPH_UNITS > 1 else ()):
                self._captured(kind, passes, p, st, capture_graph)

    def replay(self, plan, st, capture_graph, last=False):
        """One captured unit per entry of `plan` (its MFMA passes); `last`: the final entry is the pair's final unit."""
        plan = list(plan)
        tail = plan.pop() if last else None
        step = GRAPH_UNITS if GRAPH_UNITS % 2 == 0 or not self.front else 0      # (an odd run of units would end on the other parity)
        if GRAPH_UNITS <= 1:
            step = 0
        i = 0
        while i < len(plan):
            p = plan[i]
            run = 1
            while i + run < len(plan) and plan[i + run] == p:
                run += 1
            n = run
            while step and n >= step:
                self._captured("n", p, self.par, st, capture_graph).replay()
                n -= step
            for _ in range(n):
                self._captured("one", p, self.par, st, capture_graph).replay()
                if self.front:
                    self.par ^= 1
            i += run
        if last:
            self._captured("last", tail, self.par, st, capture_graph).replay()
            self.par = 0                 # (the final unit leaves the coordinate in st["coords1"])
        if self.graph is None:           # (an empty plan: nothing captured, but "a unit has run eagerly" must not repeat)
            self.graph, self.graph_last, self.graph_n = {}, {}, {}

    _tail_channels = 2                   # flow (x, y) behind the 126 motion features (core/update.py:85)

    # ---- activation scales ----------------------------------------------------------------------------------------
    def _scaled(self):
        """C8S tensors with a scale of their own (the r*h scratch shares its level's state scale)."""
        return [*self.hc8, self.up1, self.mf, self.pool0, self.up2, self.pool1, self.cor, self.flo, self.cf]

    def _state(self, st):
        """Tensors a unit changes (restored after the trial unit)."""
        return [*st["net"], st["coords1"], st["flow"]]

    def _maxima_dev(self, acts=None):
        vals = []
        for a in (self._scaled() if acts is None else acts):
            body, tail = a.absmax()
            vals += [body, tail if tail is not None else body.new_zeros(())]
        return torch.stack(vals)

    def _maxima(self, acts=None):
        return self._maxima_dev(acts).cpu().tolist()

    def _rescale(self, m):
        """New scales from the maxima `m` (scaled units, _maxima order).  Returns True when a value overflowed fp16 (scale far
        too large: the caller runs another trial with the reduced scales)."""
        overflow = False

        def pick(scale, v):
            nonlocal overflow
            if not math.isfinite(v):
                overflow = True
                return scale * 2.0 ** -12
            if v <= 0.0:
                return scale
            return scale * 2.0 ** (SCALE_EXP - math.floor(math.log2(v)))

        for i, a in enumerate(self._scaled()):
            a.scale = pick(a.scale, m[2 * i])
            a.tail_scale = pick(a.tail_scale, m[2 * i + 1]) if a.tail else a.scale
        for lvl in range(3):
            self.rh[lvl].scale = self.rh[lvl].tail_scale = self.hc8[lvl].scale
        return overflow

    def calibrate(self, st, iters):
        """Picks every C8S tensor's scale from a trial run of the loop on the pair at hand -- all `iters` units: the flow
        features grow with the disparity the iterations find -- and restores the state afterwards, so that no caller has to
        run conv.calibrate() and no layer silently computes from operands below fp16's normal range.  Once per (shape,
        weights) state; weights are re-packed with the new scales on their next use."""
        keep = [t.clone() for t in self._state(st)]
        for _ in range(4):
            self.prologue(st)
            m = None
            for k in range(max(1, iters)):
                self.unit(st, last=(k + 1 == max(1, iters)))
                cur = self._maxima_dev()
                m = cur if m is None else torch.maximum(m, cur)        # (NaN / Inf propagate: an overflow is not missed)
            overflow = self._rescale(m.cpu().tolist())
            for t, k in zip(self._state(st), keep):
                t.copy_(k)
            if not overflow:
                break
        self.calibrated = True
        self.calibrations += 1
        self.graph = self.graph_n = self.graph_last = None     # (a captured unit bakes the scales in)
        self.pinned = []
        self.par = 0

    def take_error(self):
        """True when a fused ConvGRU launch of this loop gave up waiting for a neighbour tile's flag (csrc/gru_c8.hip: the
        block then continued on stale r*h, so the results since are wrong).  One host synchronisation; the word is cleared, so a
        later forward is not blamed for this one.  (The checked forwards use `status`, which reads the same word.)"""
        return bool(self.take_error_word())

    def take_error_word(self):
        """The error word itself (and clears it): bit 0 = a fused ConvGRU launch timed out, bit 1 = a chain launch did."""
        word = int(self.err.item())
        if word:
            self.err.zero_()
        return word

    def _range_acts(self):
        """The C8S tensors whose magnitude follows the input (the state-like ones are bounded by 1)."""
        return [self.cor, self.flo, self.cf, self.mf]

    def status(self, result=None):
        """Everything a checked forward wants to know about the pair it has just computed, for ONE host synchronisation
        (dkt_loop_status: one launch over the four input-following C8S tensors and `result`, one 40-byte copy) -- round 5 paid
        three synchronisations and ten torch reductions for the same facts.  Returns LoopStatus(err, finite, ranges_ok, maxima):
        err = the error word (cleared), finite = `result` holds no Inf / NaN, maxima = (body, tail) per tensor of _range_acts in
        scaled units (Inf / NaN where a scale overflowed fp16), ranges_ok = every non-zero maximum in [2^RANGE_LO, 2^RANGE_HI)."""
        import numpy as np
        acts = self._range_acts()
        n = len(acts)
        jobs = (_ffi.C8RangeJob * n)()
        for j, a in enumerate(acts):
            jobs[j].t, jobs[j].bstride_bytes = a.data_ptr(), a.bstride_bytes
            jobs[j].B, jobs[j].C, jobs[j].H, jobs[j].W, jobs[j].tail = a.B, a.C, a.H, a.W, a.tail
        dev = self.err.device
        buf = getattr(self, "_status_buf", None)
        if buf is None or buf.numel() != 2 + 2 * n:
            buf = self._status_buf = torch.zeros(2 + 2 * n, device=dev, dtype=torch.int32)
        src, count = None, 0
        if result is not None:
            t = result
            if not t.is_contiguous():                      # (flow_up is the x plane of the up-sampled flow: check the whole tensor)
                t = t._base if (t._base is not None and t._base.is_contiguous()) else t.contiguous()
            if t.dtype != torch.float32:
                t = t.float()
            src, count = t.data_ptr(), t.numel()
        rc = _ffi.lib().dkt_loop_status(jobs, n, src, count, self.err.data_ptr(), buf.data_ptr(), _ffi.device_of(self.err),
                                        _ffi.stream_of(self.err))
        _ffi.check(rc, "dkt_loop_status")
        words = buf.cpu().numpy()                          # the forward's one host synchronisation
        maxima = words[2:].astype(np.uint16).view(np.float16).astype(np.float32)
        ok = True
        if AUTOSCALE:
            for v in maxima.tolist():
                if v != 0.0 and not (2.0 ** RANGE_LO <= v < 2.0 ** RANGE_HI):
                    ok = False
        return LoopStatus(int(words[0]), not bool(words[1]), ok, maxima.reshape(n, 2))

    def rescale_from(self, maxima):
        """New scales for the input-following tensors from the maxima a finished pair left behind (status().maxima, scaled
        units, all finite): what a recalibration needs when nothing overflowed -- no trial run (round 5 repeated the whole
        loop eagerly, up to four times, before the pair could be computed again).  Captured units bake scales in: dropped."""
        for a, (body, tail) in zip(self._range_acts(), maxima.tolist()):
            if body > 0.0 and math.isfinite(body):
                a.scale = a.scale * 2.0 ** (SCALE_EXP - math.floor(math.log2(body)))
            if a.tail:
                if tail > 0.0 and math.isfinite(tail):
                    a.tail_scale = a.tail_scale * 2.0 ** (SCALE_EXP - math.floor(math.log2(tail)))
            else:
                a.tail_scale = a.scale
        self.graph = self.graph_n = self.graph_last = None
        self.pinned = []
        self.par = 0

    def disable_fused_gru(self):
        """The two-launch form of the finest ConvGRU from here on (after a flag time-out: this device does not keep the
        launch's blocks resident, e.g. another process holds part of the CUs); captured units are dropped."""
        self.fuse_gru = False
        self.disable_chains()            # (the chain launches wait on flags the same way)

    def disable_chains(self):
        """No chain launches from here on (a chain's own time-out -- bit 1 of the error word -- leaves the fused ConvGRU
        launch alone: ADVICE r05); captured units are dropped."""
        self.chain_ok = False
        self.graph = self.graph_n = self.graph_last = None
        self.pinned = []
        self.par = 0

    def on_error_word(self, word):
        """Falls back from whichever flag-synchronised form raised `word` (take_error_word / status().err)."""
        if word & 1:
            self.disable_fused_gru()
        elif word & 2:
            self.disable_chains()

    def ranges_ok(self):
        """False when a tensor's maximum has left [2^RANGE_LO, 2^RANGE_HI) under its scale (another kind of input than the one
        calibrated on): the caller recalibrates and repeats the forward."""
        if not AUTOSCALE:
            return True
        # (the state-like tensors are bounded by 1: only the motion encoder's tensors follow the input's magnitudes)
        for v in self._maxima([self.cor, self.flo, self.cf, self.mf]):
            if v != 0.0 and not (2.0 ** RANGE_LO <= v < 2.0 ** RANGE_HI):
                return False
        return True

    @staticmethod
    def _gru_modules(ub):
        return ub.gru08, ub.gru16, ub.gru32

    # ---- pieces -------------------------------------------------------------------------------------------------
    def _chain_flags(self, key, d0, cfg0, nprob):
        f = self.chain_flags.get(key)
        if f is None:
            f = self.chain_flags[key] = c8.chain_flags(d0, cfg0, nprob, self.err.device)
        return f

    def _gru(self, lvl, gru, st, xs, cfg_zr, cfg_q, chain=False):
        h = st["net"][lvl]
        cz, cr, cq = st["inp"][lvl]
        if chain and CHAIN and self.chain_ok and cfg_zr == 4 and cfg_q == 4:
            # z|r + gates -> q + state update as ONE launch: a q tile waits for the z|r tiles of its 3x3 neighbourhood (all four
            # channel blocks: they have read every patch of the old state by then, so the in-place update is safe)
            z = self.chain_z.get(lvl)
            if z is None:
                z = self.chain_z[lvl] = torch.empty_like(h)
            d0 = c8.desc([self.hc8[lvl], *xs], gru._merged_zr(), out=z, epilogue=1, e0=cz, e1=cr, h=h, out2_c8=self.rh[lvl])
            d1 = c8.desc([self.rh[lvl], *xs], gru.convq, out=h, out_c8=self.hc8[lvl], epilogue=2, e0=cq, e1=z, h=h)
            if c8.launch_chain(d0, None, 4, d1, 4, self._chain_flags(("gru", lvl), d0, 4, 1), h, err=self.err,
                               max_blocks=CHAIN_BLOCKS, timing_only=(CHAIN == 2)):
                return
        z = c8.gate_zr([self.hc8[lvl], *xs], gru._merged_zr(), cz, cr, h, rh_c8=self.rh[lvl], cfg=cfg_zr)
        c8.gate_out([self.rh[lvl], *xs], gru.convq, cq, z, h, h, out_c8=self.hc8[lvl], cfg=cfg_q)

    def _gru_desc(self, lvl, gru, st, xs):
        cz, cr, cq = st["inp"][lvl]
        return c8.gru_desc(gru, self.hc8[lvl], xs, self.rh[lvl], cz, cr, cq, st["net"][lvl], self.gflags[lvl])

    def _gru_pair(self, st):
        """gru08 of this iteration and gru32 of the next one: ONE launch (csrc/gru_c8.hip), or two shared launches
        (z|r + gates, q + state update) where the fused form does not apply."""
        fine, _, coarse = self.grus
        if self.fuse_gru:
            d0 = self._gru_desc(0, fine, st, [self.mf, self.up1])
            d1 = self._gru_desc(2, coarse, st, [self.pool1])
            if c8.gru_launch(d0, d1, err=self.err):
                return
            self.fuse_gru = False          # this device / shape cannot hold the launch: the two-launch form from here on
        ds = []
        zs = []
        for lvl, gru, xs in ((0, fine, [self.mf, self.up1]), (2, coarse, [self.pool1])):
            h = st["net"][lvl]
            cz, cr, _ = st["inp"][lvl]
            z = torch.empty_like(h)
            zs.append(z)
            ds.append(c8.desc([self.hc8[lvl], *xs], gru._merged_zr(), out=z, epilogue=1, e0=cz, e1=cr, h=h, out2_c8=self.rh[lvl]))
        c8.launch_pair(ds[0], ds[1], zs[0], self.cfg["zr08"])
        ds = []
        for (lvl, gru, xs), z in zip(((0, fine, [self.mf, self.up1]), (2, coarse, [self.pool1])), zs):
            h = st["net"][lvl]
            cq = st["inp"][lvl][2]
            ds.append(c8.desc([self.rh[lvl], *xs], gru.convq, out=h, out_c8=self.hc8[lvl], epilogue=2, e0=cq, e1=z, h=h))
        c8.launch_pair(ds[0], ds[1], zs[0], self.cfg["q08"])

    def _fork_stem(self, ref, fn):
        """The motion encoder's 7x7 stem beside the lookup (independent until convc2 | convf2 joins them): a third stream."""
        if FORK < 2:
            fn()
            return lambda: None
        from .update import _side_stream
        main = torch.cuda.current_stream(ref.device)
        side = _side_stream(ref.device, slot=1)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            fn()
        return lambda: main.wait_stream(side)

    def _motion_front(self, st):
        """convf1 on the flow and the lookup + convc1 (core/update.py:76-77): two launches behind the head's own finish."""
        enc = self.ub.encoder
        join = self._fork_stem(st["flow"], lambda: c8.stem7_c8(st["flow"], enc.convf1, self.flo))
        if st["corr"].lookup_conv1x1(st["coords1"], enc.convc1, out_c8=self.cor) is None:
            cor = st["corr"].lookup_conv1x1(st["coords1"], enc.convc1)
            if cor is None:       # (row-layout pyramid, other level counts / radii: the lookup tensor, then the 1x1 layer)
                cor = _conv.conv2d(st["corr"](st["coords1"]), enc.convc1, relu=True)
            c8.pack(cor, self.cor)
        join()

    def _motion_tail(self, st, chain=False):
        enc = self.ub.encoder
        d0 = c8.desc([self.cor], enc.convc2, relu=True, out_c8=self.cf, out_c8_ch0=0)
        d1 = c8.desc([self.flo], enc.convf2, relu=True, out_c8=self.cf, out_c8_ch0=64)
        if chain and CHAIN and self.chain_ok and self.cfg["c2"] == 4 and self.cfg["enc"] == 3:
            d2 = c8.desc([self.cf], enc.conv, relu=True, out_c8=self.mf, tail=st["flow"])
            if c8.launch_chain(d0, d1, 4, d2, 3, self._chain_flags("enc", d0, 4, 2), st["flow"], err=self.err,
                               max_blocks=CHAIN_BLOCKS, timing_only=(CHAIN == 2)):
                return
        c8.launch_pair(d0, d1, st["flow"], self.cfg["c2"])
        c8.conv2d_c8([self.cf], enc.conv, relu=True, out_c8=self.mf, tail=st["flow"], cfg=self.cfg["enc"])

    def _motion(self, st):
        self._motion_front(st)
        self._motion_tail(st)

    def _coords(self, st):
        """The two x-coordinate buffers of the fused front: st["coords1"][:, :1] and a twin."""
        if self.cx is None or self.cx[0].data_ptr() != st["coords1"].data_ptr():
            self.cx = [st["coords1"][:, :1], torch.empty_like(st["coords1"][:, :1].contiguous())]
        return self.cx

    def _head(self, st, front=False):
        """Flow head (core/update.py:6-14) and the coordinate update (raft_stereo.py:165-168).  `front`: the update, the
        lookup + convc1 and convf1 of the NEXT iteration's motion encoder in the head's second launch (dkt_motion_front_c8):
        reads the coordinate from cx[par], writes cx[1 - par]."""
        fh = self.ub.flow_head
        if front:
            cx = self._coords(st)
            planes, n_co = c8.head_planes([self.hc8[0]], fh.conv1, _leading_outputs(fh.conv2, 1), cfg=self.cfg["head"])
            enc = self.ub.encoder
            c8.motion_front(st["corr"], planes, n_co, _leading_outputs(fh.conv2, 1).bias, cx[self.par], cx[1 - self.par],
                            st["coords0"][:, :1], st["flow"], enc.convc1, self.cor, enc.convf1, self.flo)
            self.par ^= 1
            return
        target = self._coords(st)[self.par] if self.front else st["coords1"][:, :1]
        if FUSE_HEAD:
            # conv2 (x output only: raft_stereo.py:165) from per-tap projections made in conv1's epilogue
            c8.head([self.hc8[0]], fh.conv1, _leading_outputs(fh.conv2, 1), target,
                    diff=(st["coords0"][:, :1], st["flow"][:, :1]), cfg=self.cfg["head"])
        else:
            c8.conv2d_c8([self.hc8[0]], fh.conv1, relu=True, out=self.hidden, cfg=self.cfg["head"])
            _conv.conv2d_accumulate(self.hidden, _leading_outputs(fh.conv2, 1), target,
                                    diff=(st["coords0"][:, :1], st["flow"][:, :1]))
        if self.front and self.par:
            self.cx[0].copy_(self.cx[1])          # the pair's final coordinate lives in st["coords1"]
            self.par = 0

    def _mask(self, st):
        """0.25 * mask(net[0]) (core/update.py:107-110, :136): the 3x3 layer reads the C8S twin of the final hidden state, the
        factor is folded into the 1x1 layer's weights (a power of two: the same bits)."""
        mh = self.mask_head
        if mh is None:
            self.mask_out = None
            return
        hid = c8.conv2d_c8([self.hc8[0]], mh[0], relu=True, cfg=1)
        self.mask_out = _conv.conv2d(hid, _scaled_layer(mh[2], 0.25), out=self.mask_out)

    def _mid(self, st, chain=False):
        n0, n1, n2 = st["net"]
        if PAIR_RESAMPLE:
            c8.resample_pair_c8(("pool", n0, self.pool0), ("interp", n2, self.up2))
        else:
            c8.pool2x_c8(n0, self.pool0)
            c8.interp_c8(n2, self.up2)
        self._gru(1, self.grus[1], st, [self.pool0, self.up2], self.cfg["zr16"], self.cfg["q16"], chain=chain)
        if PAIR_RESAMPLE:
            c8.resample_pair_c8(("interp", n1, self.up1), ("pool", n1, self.pool1))
        else:
            c8.interp_c8(n1, self.up1)
            c8.pool2x_c8(n1, self.pool1)

    def unit(self, st, last=False):
        """Iteration i from the finest GRU on, and iteration i + 1 up to it: the middle GRU's chain of the NEXT iteration runs
        on a second stream beside this iteration's head and the next motion encoder (round 4: these launches are a few
        dozen microseconds each, half of it fixed cost -- unlike the wide ones they do overlap: 35.0 -> 33.0 ms per pair with
        the fork alone).  `last`: the pair's final iteration (nothing of a next one is started)."""
        self._gru_pair(st)
        if last:
            # the up-sampling mask head reads the final hidden state only: beside the flow head, on the forked stream
            if FORK and self.mask_head is not None:
                from .update import _side_stream
                dev = st["net"][0].device
                main = torch.cuda.current_stream(dev)
                side = _side_stream(dev, slot=0)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    self._mask(st)
                self._head(st)
                main.wait_stream(side)
            else:
                self._head(st)
                self._mask(st)
            return
        front = self.front and FUSE_HEAD
        if FORK:
            from .update import _side_stream
            dev = st["net"][0].device
            main = torch.cuda.current_stream(dev)
            side = _side_stream(dev, slot=0)
            if HEAD_FIRST:
                # the head is enqueued BEFORE the forked chain: in the instantiated graph the fused GRU launch's first
                # successor stays on its queue, so the critical chain gru -> head -> motion encoder -> gru crosses no queue
                # boundary (a cross-queue dependency costs ~10 us of idle device, twice per iteration)
                forked = torch.cuda.Event()
                forked.record(main)
                self._head(st, front)
                side.wait_event(forked)
            else:
                side.wait_stream(main)
            with torch.cuda.stream(side):
                self._mid(st, chain=True)
            if not HEAD_FIRST:
                self._head(st, front)
            if not front:
                self._motion_front(st)
            self._motion_tail(st, chain=True)
            main.wait_stream(side)
        else:
            self._head(st, front)
            self._mid(st, chain=True)
            if not front:
                self._motion_front(st)
            self._motion_tail(st, chain=True)

    def prologue(self, st):
        """Per pair: hidden states into their C8S twins, the coarsest and the middle GRU of iteration 0 and its motion features."""
        n0, n1, n2 = st["net"]
        self.par = 0
        for lvl, n in enumerate(st["net"]):
            c8.pack(n, self.hc8[lvl])
        if FORK and PROLOGUE_FORK:
            # the motion features of iteration 0 need the correlation volume and the initial coordinates only: beside the
            # hidden-state chain (pack, pool, coarsest and middle GRU) instead of behind it
            from .update import _side_stream
            main = torch.cuda.current_stream(n0.device)
            side = _side_stream(n0.device, slot=0)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self._motion(st)
        c8.pool2x_c8(n1, self.pool1)
        self._gru(2, self.grus[2], st, [self.pool1], 4, 4)
        self._mid(st)
        if FORK and PROLOGUE_FORK:
            main.wait_stream(side)
        else:
            self._motion(st)


class C8LoopIGEV(C8Loop):
    """The IGEV refinement loop (meta_arch/igev_stereo/igev_stereo.py:199-210 with the update block of
    igev_stereo/update.py:94-142) on the same kernels: gru04 / gru08 / gru16 at 1/4, 1/8, 1/16, the geometry-encoding
    lookup (geometry.py:29-69) in front of the 162 -> 64 1x1 layer, a 1-channel disparity stem, the 127 + 1 channel motion
    features and the one-output disparity head whose result is added to the running disparity in dkt_head_finish.
    ``st``: dict(net, inp, disp, coords, geo_fn)."""

    _tail_channels = 1                   # the running disparity behind the 127 motion features (igev_stereo/update.py:92)

    def _front_supported(self, st):
        return False                     # (the geometry lookup has its own fused kernel, geo_feat.hip)

    @staticmethod
    def _gru_modules(ub):
        return ub.gru04, ub.gru08, ub.gru16

    def _state(self, st):
        return [*st["net"], st["disp"]]

    def _motion_front(self, st):
        enc = self.ub.encoder
        # geometry lookup + convc1 + ReLU in one kernel, straight into the C8S operand (dkt_geo_lookup_conv1x1)
        join = self._fork_stem(st["disp"], lambda: c8.stem7_c8(st["disp"], enc.convd1, self.flo))
        if st["geo_fn"].lookup_conv1x1(st["disp"], st["coords"], enc.convc1, out_c8=self.cor) is None:
            geo = st["geo_fn"](st["disp"], st["coords"])
            c8.pack(_conv.conv2d(geo, enc.convc1, relu=True), self.cor)
        join()

    def _motion_tail(self, st, chain=False):
        enc = self.ub.encoder
        d0 = c8.desc([self.cor], enc.convc2, relu=True, out_c8=self.cf, out_c8_ch0=0)
        d1 = c8.desc([self.flo], enc.convd2, relu=True, out_c8=self.cf, out_c8_ch0=64)
        if chain and CHAIN and self.chain_ok and self.cfg["c2"] == 4 and self.cfg["enc"] == 3:
            d2 = c8.desc([self.cf], enc.conv, relu=True, out_c8=self.mf, tail=st["disp"])
            if c8.launch_chain(d0, d1, 4, d2, 3, self._chain_flags("enc", d0, 4, 2), st["disp"], err=self.err,
                               max_blocks=CHAIN_BLOCKS, timing_only=(CHAIN == 2)):
                return
        c8.launch_pair(d0, d1, st["disp"], self.cfg["c2"])
        c8.conv2d_c8([self.cf], enc.conv, relu=True, out_c8=self.mf, tail=st["disp"], cfg=self.cfg["enc"])

    def _head(self, st, front=False):
        dh = self.ub.disp_head
        c8.head([self.hc8[0]], dh.conv1, dh.conv2, st["disp"], cfg=self.cfg["head"])      # disp += delta (igev_stereo.py:209)
