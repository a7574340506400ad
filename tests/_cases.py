"""Parity cases: one definition shared by make_golden.py (which runs the imported
reference on them and stores its outputs) and by the tests (which run the
oracle / the HIP path on the same regenerated inputs)."""
import numpy as np

import _synth

# ---- RAFT correlation build + lookup (core/corr.py:110-156) -----------------
CORR_CASES = {
    "small":  dict(seed=11, B=2, C=32, H=5, W=40, W2=40, L=4, r=4, hard=False),
    "odd":    dict(seed=12, B=1, C=24, H=4, W=39, W2=39, L=3, r=4, hard=True),
    "r3":     dict(seed=13, B=1, C=17, H=3, W=64, W2=64, L=2, r=3, hard=True),
    "w1neW2": dict(seed=14, B=1, C=16, H=3, W=40, W2=52, L=3, r=4, hard=False),
    "kitti":  dict(seed=15, B=1, C=256, H=2, W=312, W2=312, L=4, r=4, hard=False),
    "tiny":   dict(seed=16, B=1, C=4, H=2, W=9, W2=9, L=1, r=1, hard=True),
}


def corr_inputs(c):
    f1, f2 = _synth.fmap_pair(c["seed"], c["B"], c["C"], c["H"], c["W"], c["W2"])
    mk = _synth.coords_hard if c["hard"] else _synth.coords
    co = mk(c["seed"], c["B"], c["H"], c["W"])
    return f1, f2, co


# ---- IGEV geometry volume (meta_arch/igev_stereo/geometry.py) ------------------
GEO_CASES = {
    "small": dict(seed=21, B=1, Cm=24, C=8, D=16, H=4, W=40, L=2, r=4),
    "b2":    dict(seed=22, B=2, Cm=16, C=8, D=12, H=3, W=25, L=2, r=4),
    "l3":    dict(seed=23, B=1, Cm=8, C=4, D=24, H=2, W=48, L=3, r=2),
}


def geo_inputs(c):
    m1, m2 = _synth.fmap_pair(c["seed"], c["B"], c["Cm"], c["H"], c["W"])
    geo = _synth.normal((c["B"], c["C"], c["D"], c["H"], c["W"]), c["seed"], "geo")
    disp = _synth.uniform((c["B"], 1, c["H"], c["W"]), -2.0, float(c["D"]) + 2.0, c["seed"], "disp")
    disp[..., 0::5] = np.round(disp[..., 0::5])
    coords = np.broadcast_to(np.arange(c["W"], dtype=np.float32).reshape(1, 1, c["W"], 1),
                             (c["B"], c["H"], c["W"], 1)).copy()
    return m1, m2, geo, disp, coords


# ---- PCVNet correlation block (meta_arch/pcvnet/corr.py) -----------------------
PCV_CASES = {
    "ds2":   dict(seed=81, B=1, C=16, H=3, W=64, L=3, S=9, G=2, downsample=2),    # factor 4
    "ds3":   dict(seed=82, B=2, C=8, H=2, W=40, L=4, S=5, G=3, downsample=3),     # factor 2
    "odd":   dict(seed=83, B=1, C=12, H=2, W=77, L=3, S=9, G=1, downsample=2),    # widths 77, 19, 4
}


def pcv_inputs(c):
    f1, f2 = _synth.fmap_pair(c["seed"], c["B"], c["C"], c["H"], c["W"])
    base = np.arange(c["W"], dtype=np.float32).reshape(1, 1, 1, c["W"])
    shp = (c["B"], c["G"], c["H"], c["W"])
    coords = (base - _synth.uniform(shp, 0.0, 0.6 * c["W"], c["seed"], "pcvc")).astype(np.float32)
    coords[..., 0::7] = np.round(coords[..., 0::7])
    sigma = _synth.uniform(shp, 0.25, 3.0, c["seed"], "pcvs")
    sigma[..., 0::5] = 1.0
    return f1, f2, coords, sigma


# ---- CGI normalised correlation volumes (meta_arch/cgi/submodule.py:143-180) --------
NORMCORR_CASES = {
    "cgi":  dict(seed=91, B=2, C=12, H=3, W=24, D=8, G=1),
    "g4":   dict(seed=92, B=1, C=32, H=2, W=30, D=12, G=4),
    "dgtw": dict(seed=93, B=1, C=6, H=2, W=5, D=7, G=2),
    "wide": dict(seed=94, B=1, C=48, H=2, W=40, D=12, G=1),      # one group wider than 16 channels (CGI: cpg = C)
}


# ---- up-sampling either side of the loop (raft_stereo.py:70-82, igev_stereo/submodule.py:242-254) ----
UPSAMPLE_CASES = {
    "f4":  dict(seed=101, N=2, D=2, H=5, W=7, nd=2),
    "f2":  dict(seed=102, N=1, D=1, H=3, W=9, nd=1),
    "f8":  dict(seed=103, N=1, D=2, H=2, W=3, nd=3),
}
CONTEXT_UP_CASES = {
    "small": dict(seed=111, B=2, h=4, w=6),
    "one":   dict(seed=112, B=1, h=1, w=5),
}


def upsample_inputs(c):
    f = 2 ** c["nd"]
    flow = _synth.normal((c["N"], c["D"], c["H"], c["W"]), c["seed"], "flow", scale=5.0)
    mask = _synth.normal((c["N"], 9 * f * f, c["H"], c["W"]), c["seed"], "mask", scale=2.0)
    return flow, mask, f


def context_up_inputs(c):
    disp = _synth.uniform((c["B"], 1, c["h"], c["w"]), 0.0, 48.0, c["seed"], "disp")
    wts = _synth.normal((c["B"], 9, 4 * c["h"], 4 * c["w"]), c["seed"], "wts")
    e = np.exp(wts - wts.max(axis=1, keepdims=True))
    return disp, (e / e.sum(axis=1, keepdims=True)).astype(np.float32)


# ---- cost volumes ----------------------------------------------------------------
GWC_CASES = {
    "igev":  dict(seed=31, B=2, C=96, H=3, W=24, D=8, G=8),      # cpg 12
    "gwc":   dict(seed=32, B=1, C=320, H=2, W=30, D=12, G=40),   # cpg 8
    "dgtw":  dict(seed=33, B=1, C=8, H=2, W=6, D=9, G=2),        # D > W: empty slices
    "g1":    dict(seed=34, B=1, C=5, H=2, W=17, D=4, G=1),
    "cpg40": dict(seed=35, B=1, C=80, H=2, W=21, D=6, G=2),      # groups wider than 16 channels
}
CONCAT_CASES = {
    "gc":   dict(seed=41, B=2, C=12, H=3, W=24, D=8),
    "dgtw": dict(seed=42, B=1, C=3, H=2, W=5, D=7),
}


def volume_inputs(c):
    return _synth.fmap_pair(c["seed"], c["B"], c["C"], c["H"], c["W"])


# ---- update operator -----------------------------------------------------------------
GRU_CASES = {
    "small": dict(seed=51, B=2, hidden=16, inputs=(24,), H=6, W=10),
    "two_x": dict(seed=52, B=1, hidden=128, inputs=(128, 128), H=5, W=7),
}


def gru_inputs(c):
    s = c["seed"]
    shp = (c["B"], c["hidden"], c["H"], c["W"])
    h = np.tanh(_synth.normal(shp, s, "h"))
    czrq = _synth.normal((c["B"], 3 * c["hidden"], c["H"], c["W"]), s, "czrq", scale=0.5)
    xs = [_synth.normal((c["B"], ci, c["H"], c["W"]), s, "x%d" % i) for i, ci in enumerate(c["inputs"])]
    return h, czrq, xs


UPDATE_CASES = {
    # name: (flavour, n_gru_layers, slow_fast, H, W at the finest scale)
    "raft3": dict(seed=61, igev=False, n=3, B=1, H=8, W=12),
    "raft2": dict(seed=62, igev=False, n=2, B=2, H=8, W=12),
    "raft1": dict(seed=63, igev=False, n=1, B=1, H=8, W=12),
    "igev3": dict(seed=64, igev=True, n=3, B=1, H=8, W=12),
}


def update_cfg(c):
    cfg = dict(corr_levels=2 if c["igev"] else 4, corr_radius=4, n_downsample=2,
               n_gru_layers=c["n"], hidden_dims=[128, 128, 128], slow_fast_gru=False)
    return cfg


def update_inputs(c):
    s, B, H, W = c["seed"], c["B"], c["H"], c["W"]
    cfg = update_cfg(c)
    K = 2 * cfg["corr_radius"] + 1
    cor_planes = cfg["corr_levels"] * K * (9 if c["igev"] else 1)
    net, inp = [], []
    for i in range(3):
        hh, ww = H >> i, W >> i
        net.append(np.tanh(_synth.normal((B, 128, hh, ww), s, "net%d" % i)))
        inp.append(_synth.normal((B, 384, hh, ww), s, "inp%d" % i, scale=0.5))
    corr = _synth.normal((B, cor_planes, H, W), s, "corr")
    flow = _synth.normal((B, 1 if c["igev"] else 2, H, W), s, "flow", scale=3.0)
    return net, inp, corr, flow


# ---- end to end ------------------------------------------------------------------------
E2E_CASES = {
    "64x128_it4":   dict(seed=0, B=1, H=64, W=128, iters=4, shift=12, stride=1),
    "64x128_it12":  dict(seed=1, B=1, H=64, W=128, iters=12, shift=12, stride=1),
    "256x512_it8":  dict(seed=0, B=1, H=256, W=512, iters=8, shift=12, stride=4),   # BASELINE cfg 1
    "256x512_it32": dict(seed=2, B=1, H=256, W=512, iters=32, shift=40, stride=4),
    "736x1248_it32": dict(seed=3, B=1, H=736, W=1248, iters=32, shift=40, stride=8),  # BASELINE cfg 2: the benchmark workload
}
E2E_WEIGHT_SEED = 7

# slow-fast GRU schedule (raft_stereo.py:156-159): extra coarse / mid updates before every full update
E2E_SLOWFAST_CASES = {
    "sf3_64x128_it6": dict(seed=4, B=1, H=64, W=128, iters=6, shift=12, n=3),
    "sf2_64x128_it6": dict(seed=5, B=1, H=64, W=128, iters=6, shift=12, n=2),
}

# the non-default backbones of raft_stereo.py:43-54, 97-108 (round 5): the context encoder's trunk shared by both images,
# and the correlation on the down-sampled images themselves (3 feature channels)
E2E_BACKBONE_CASES = {
    "shared_64x128_it6":      dict(seed=6, B=1, H=64, W=128, iters=6, shift=12, over=dict(shared_backbone=True)),
    "interpolate_64x128_it6": dict(seed=8, B=2, H=64, W=128, iters=6, shift=12, over=dict(backbone_type="interpolate")),
}

IGEV_LOOP_CASES = {
    "small": dict(seed=71, B=1, Cm=24, C=8, D=16, H=8, W=16, iters=3, n=3, slow_fast=False, stride=1),
    # slow-fast schedule of igev_stereo.py:204-207, 3 and 2 GRU layers
    "sf3":   dict(seed=73, B=1, Cm=24, C=8, D=16, H=8, W=16, iters=4, n=3, slow_fast=True, stride=1),
    "sf2":   dict(seed=74, B=1, Cm=24, C=8, D=16, H=8, W=16, iters=4, n=2, slow_fast=True, stride=1),
    # BASELINE cfg 3 shapes (736x1248 -> 184x312, 96 match channels, geometry volume 8 x 48), 32 iterations
    "kitti": dict(seed=72, B=1, Cm=96, C=8, D=48, H=184, W=312, iters=32, n=3, slow_fast=False, stride=4),
}


def igev_loop_cfg(c):
    return dict(corr_levels=2, corr_radius=4, n_downsample=2, n_gru_layers=c["n"],
                hidden_dims=[128, 128, 128], slow_fast_gru=c["slow_fast"])


def igev_loop_inputs(c):
    """match features, geometry volume, initial disparity, coords, hidden states, context inputs."""
    s, B, H, W = c["seed"], c["B"], c["H"], c["W"]
    m1, m2, geo, disp, coords = geo_inputs(dict(c, L=2, r=4))
    net = [np.tanh(_synth.normal((B, 128, H >> i, W >> i), s, "net%d" % i)) for i in range(3)]
    inp = [_synth.normal((B, 384, H >> i, W >> i), s, "inp%d" % i, scale=0.5) for i in range(3)]
    return m1, m2, geo, np.abs(disp), coords, net, inp


# ---- GwcNet end to end (meta_arch/gwcnet/gwc_main.py:279-326, eval / test_mode) -----------------
GWCNET_CASES = {
    "64x128":  dict(seed=6, B=1, H=64, W=128, shift=12, stride=1),
    "96x160_b2": dict(seed=8, B=2, H=96, W=160, shift=20, stride=2),
    # BASELINE cfg 5: 540x960 padded to 544x960 (InputPadder divis_by=32), D = 192
    "544x960": dict(seed=9, B=1, H=544, W=960, shift=40, stride=8),
}
GWCNET_WEIGHT_SEED = 17


# ---- evaluation chain (tools/evaluate_stereo.py:124-134): pad to /32 -> forward -> unpad ----------
EVAL_CASES = {
    "raft_100x187_it4": dict(seed=21, H=100, W=187, shift=12, iters=4),     # neither side divisible by 32
}
