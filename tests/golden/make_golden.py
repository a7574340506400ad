#!/usr/bin/env python3
"""Generates the golden fixtures in this directory by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

For every case in tests/_cases.py the reference (jiaw-z/DKT-Stereo, imported
unmodified through _refimport.py) is run on CPU/fp32 on inputs regenerated from
seeds (tests/_synth.py); its outputs are written as small .npz files (large
outputs are strided, the stride is stored).  While it is at it the script pins
both oracles -- oracle/torch_oracle.py and oracle/dkt_oracle.c -- against the
reference and records the observed max-abs differences in MANIFEST.json; it
fails if any exceeds the bound the tests later use.

Only data is stored here: inputs' seeds, the reference's numeric outputs and
metadata.  No reference source or bytecode is copied.
"""
import json
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

import _cases  # noqa: E402
import _refimport  # noqa: E402
import _synth  # noqa: E402
from oracle import c_oracle as co  # noqa: E402
from oracle import torch_oracle as to  # noqa: E402

torch.set_num_threads(8)
T = torch.from_numpy
MANIFEST = {"reference": "jiaw-z/DKT-Stereo @ 2024_08_07", "torch": torch.__version__, "pins": {}}


def pin(name, got, want, bound):
    d = float(np.max(np.abs(np.asarray(got, np.float64) - np.asarray(want, np.float64)))) if np.size(want) else 0.0
    MANIFEST["pins"][name] = {"max_abs": d, "bound": bound}
    status = "ok" if d <= bound else "FAIL"
    print("  pin %-46s max_abs %.3e (bound %.1e) %s" % (name, d, bound, status))
    if not d <= bound:
        raise SystemExit("oracle disagrees with the reference: " + name)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("  wrote %s (%.1f KB)" % (os.path.basename(path), os.path.getsize(path) / 1024))


def flat(p):
    return p.reshape(p.shape[0], -1)


@torch.no_grad()
def gen_corr(ref):
    print("corr build / lookup")
    out = {}
    for name, c in _cases.CORR_CASES.items():
        f1, f2, coords = _cases.corr_inputs(c)
        blk = ref.corr.CorrBlock1D(T(f1), T(f2), num_levels=c["L"], radius=c["r"])
        pyr = [flat(p.numpy()) for p in blk.corr_pyramid[:c["L"]]]
        look = blk(T(coords)).numpy()
        alt = ref.corr.PytorchAlternateCorrBlock1D(T(f1), T(f2), num_levels=c["L"], radius=c["r"])(T(coords)).numpy()
        cosb = ref.corr.CorrBlock1D_Cosine(T(f1), T(f2), num_levels=c["L"], radius=c["r"])
        cos0 = flat(cosb.corr_pyramid[0].numpy())
        cosl = cosb(T(coords)).numpy()
        stride = 1 if pyr[0].size < 40000 else 13
        for i, p in enumerate(pyr):
            out["%s/pyr%d" % (name, i)] = p[::stride]
        out["%s/pyr_stride" % name] = np.int64(stride)
        out["%s/lookup" % name] = look
        out["%s/alt" % name] = alt
        out["%s/cos0" % name] = cos0[::stride]
        out["%s/coslookup" % name] = cosl
        # --- pin the oracles
        tp = to.corr1d_pyramid(T(f1), T(f2), c["L"])
        for i in range(c["L"]):
            pin("torch.corr_pyramid[%s][%d]" % (name, i), flat(tp[i].numpy()), pyr[i], 0.0)
        pin("torch.corr_lookup[%s]" % name, to.corr1d_lookup(tp, T(coords), c["r"]).numpy(), look, 0.0)
        pin("torch.corr_lookup_alt[%s]" % name, to.corr1d_lookup_alt(T(f1), T(f2), T(coords), c["L"], c["r"]).numpy(), alt, 0.0)
        cp = co.corr1d_build(f1, f2, c["L"])
        scale = float(np.abs(pyr[0]).max())
        for i in range(c["L"]):
            pin("c.corr_build[%s][%d]" % (name, i), cp[i], pyr[i], 4e-6 * max(scale, 1.0))
        pin("c.pool_pyramid[%s]" % name, np.concatenate([p.ravel() for p in co.pool_pyramid(pyr[0], c["L"])]),
            np.concatenate([p.ravel() for p in pyr]), 0.0)
        pin("c.corr_lookup[%s] (on reference pyramid)" % name, co.corr1d_lookup(pyr, coords, c["r"]), look, 0.0)
        pin("c.corr_lookup_alt[%s]" % name, co.corr1d_lookup_alt(f1, f2, coords, c["L"], c["r"]), alt,
            4e-6 * max(scale, 1.0))
    save("corr", **out)


@torch.no_grad()
def gen_upsample(ref):
    print("convex / context up-sampling")
    out = {}
    for name, c in _cases.UPSAMPLE_CASES.items():
        flow, mask, f = _cases.upsample_inputs(c)
        fake = SimpleNamespace(args=SimpleNamespace(n_downsample=c["nd"]))
        want = ref.RAFTStereo.upsample_flow(fake, T(flow), T(mask)).numpy()
        out["convex/%s" % name] = want
        pin("torch.convex_upsample[%s]" % name, to.convex_upsample(T(flow), T(mask), f).numpy(), want, 0.0)
        pin("c.convex_upsample[%s]" % name, co.convex_upsample(flow, mask, f), want, 2e-6 * float(np.abs(want).max()))
    for name, c in _cases.CONTEXT_UP_CASES.items():
        disp, wts = _cases.context_up_inputs(c)
        want = ref.igev_sub.context_upsample(T(disp), T(wts)).numpy()
        out["context/%s" % name] = want
        pin("torch.context_upsample[%s]" % name, to.context_upsample(T(disp), T(wts)).numpy(), want, 0.0)
        pin("c.context_upsample[%s]" % name, co.context_upsample(disp, wts), want, 2e-6 * float(np.abs(want).max()))
    save("upsample", **out)


def gen_files(ref):
    """Tiny dataset-format files + what the reference's readers return for them."""
    from PIL import Image
    print("on-disk formats (core/utils/frame_utils.py)")
    fu = _refimport.load_frame_utils()
    d = os.path.join(HERE, "files")
    os.makedirs(os.path.join(d, "disparities"), exist_ok=True)
    os.makedirs(os.path.join(d, "occlusions"), exist_ok=True)
    rng = np.random.Generator(np.random.PCG64(2024))
    out = {}
    disp = (rng.random((7, 11)) * 90).astype(np.float32)
    fu.writePFM(os.path.join(d, "disp.pfm"), disp)                          # written by the reference
    out["pfm"] = np.ascontiguousarray(fu.readPFM(os.path.join(d, "disp.pfm")))
    rgb = (rng.random((5, 6, 3)) * 3).astype(np.float32)
    with open(os.path.join(d, "color_be.pfm"), "wb") as f:                  # 3-channel, big endian
        f.write(b"PF\n6 5\n1.0\n")
        f.write(np.flipud(rgb).astype(">f4").tobytes())
    out["pfm_color"] = np.ascontiguousarray(fu.readPFM(os.path.join(d, "color_be.pfm")))
    out["read_gen_pfm_color"] = np.ascontiguousarray(fu.read_gen(os.path.join(d, "color_be.pfm")))
    uv = rng.normal(size=(6, 9, 2)).astype(np.float32)
    fu.writeFlow(os.path.join(d, "flow.flo"), uv)
    out["flo"] = fu.readFlow(os.path.join(d, "flow.flo"))
    k = (rng.random((8, 12)) * 60000).astype(np.uint16)
    k[::3, ::4] = 0
    Image.fromarray(k).save(os.path.join(d, "kitti_disp.png"))              # 16-bit grey, KITTI disp_occ_0 format
    out["kitti_disp_expected"] = k.astype(np.float64) / 256.0               # cv2.imread(ANYDEPTH)/256.0, frame_utils.py:153
    s = (rng.random((6, 7, 3)) * 255).astype(np.uint8)
    Image.fromarray(s).save(os.path.join(d, "disparities", "frame.png"))
    occ = ((rng.random((6, 7)) > 0.7) * 255).astype(np.uint8)
    Image.fromarray(occ).save(os.path.join(d, "occlusions", "frame.png"))
    sd, sv = fu.readDispSintelStereo(os.path.join(d, "disparities", "frame.png"))
    out["sintel_disp"], out["sintel_valid"] = sd, sv
    depth = (rng.random((4, 5)) * 50 + 0.5).astype(np.float32)
    np.save(os.path.join(d, "depth.npy"), depth)
    td, tv = fu.readDispTartanAir(os.path.join(d, "depth.npy"))
    out["tartan_disp"], out["tartan_valid"] = td, tv
    save("files", **out)


def gen_corr_bwd(ref):
    """Autograd of the reference through lookup, pyramid and all-pairs correlation."""
    print("corr lookup / build backward")
    out = {}
    for name, c in _cases.CORR_CASES.items():
        f1, f2, coords = _cases.corr_inputs(c)
        K = 2 * c["r"] + 1
        R = _synth.normal((c["B"], c["L"] * K, c["H"], c["W"]), c["seed"], "gout")
        a, b = T(f1).requires_grad_(True), T(f2).requires_grad_(True)
        blk = ref.corr.CorrBlock1D(a, b, num_levels=c["L"], radius=c["r"])
        look = blk(T(coords))
        levels = blk.corr_pyramid[:c["L"]]
        glv = torch.autograd.grad(look, levels, T(R), retain_graph=True, allow_unused=True)
        gf1, gf2 = torch.autograd.grad(look, [a, b], T(R))
        glv = [flat(g.numpy()) for g in glv]
        for i, g in enumerate(glv):
            out["%s/glevel%d" % (name, i)] = g
        out["%s/gf1" % name] = gf1.numpy()
        out["%s/gf2" % name] = gf2.numpy()
        # --- pin the C restatement
        widths = [c["W2"] >> i for i in range(c["L"])]
        rows = c["B"] * c["H"] * c["W"]
        cg = co.corr1d_lookup_bwd(R, coords, c["r"], widths, rows)
        # autograd's gradient of level i is TOTAL (direct taps + what flows back through the
        # pooled levels above it): compare with the pooled-back chain of the per-level scatters
        for i in range(c["L"]):
            pin("c.corr_lookup_bwd+pool_bwd[%s][%d]" % (name, i), co.corr1d_pool_bwd(cg[i:], 1.0), glv[i], 0.0)
        g0 = co.corr1d_pool_bwd(cg, float(np.sqrt(np.float32(c["C"]))))
        cf1, cf2 = co.corr1d_build_bwd(g0, f1, f2)
        s1 = max(float(np.abs(gf1.numpy()).max()), 1.0)
        pin("c.corr_build_bwd.f1[%s]" % name, cf1, gf1.numpy(), 4e-6 * s1)
        pin("c.corr_build_bwd.f2[%s]" % name, cf2, gf2.numpy(), 4e-6 * max(float(np.abs(gf2.numpy()).max()), 1.0))
    save("corr_bwd", **out)


def gen_geo_bwd(ref):
    """Autograd of the reference through Combined_Geo_Encoding_Volume (IGEV flavour)."""
    print("IGEV geometry volume backward")
    out = {}
    for name, c in _cases.GEO_CASES.items():
        m1, m2, geo, disp, coords = _cases.geo_inputs(c)
        K = 2 * c["r"] + 1
        R = _synth.normal((c["B"], c["L"] * K * (c["C"] + 1), c["H"], c["W"]), c["seed"], "ggeo")
        a, b, gv = T(m1).requires_grad_(True), T(m2).requires_grad_(True), T(geo).requires_grad_(True)
        g = ref.GeoVolume(a, b, gv, num_levels=c["L"], radius=c["r"])
        look = g(T(disp), T(coords))
        ga, gb, gg = torch.autograd.grad(look, [a, b, gv], T(R))
        out["%s/gm1" % name], out["%s/gm2" % name], out["%s/ggeo" % name] = ga.numpy(), gb.numpy(), gg.numpy()
        # --- pin the C restatement: scatter + pooled-back chains (+ fp64 contractions for the features)
        B, C, D, H, W = geo.shape
        cg, ci = co.geo_lookup_bwd(R, disp, coords, C, D, c["W"], c["L"], c["r"])
        tg = co.corr1d_pool_bwd(cg, 1.0).reshape(B, H, W, C, D).transpose(0, 3, 4, 1, 2)      # (n*C+c, D) -> (B,C,D,H,W)
        pin("c.geo_lookup_bwd+pool_bwd.geo[%s]" % name, tg, gg.numpy(), 0.0)
        t0 = co.corr1d_pool_bwd(ci, 1.0)
        cf1, cf2 = co.corr1d_build_bwd(t0, m1, m2)
        pin("c.geo_bwd.m1[%s]" % name, cf1, ga.numpy(), 4e-6 * max(float(np.abs(ga.numpy()).max()), 1.0))
        pin("c.geo_bwd.m2[%s]" % name, cf2, gb.numpy(), 4e-6 * max(float(np.abs(gb.numpy()).max()), 1.0))
    save("geo_bwd", **out)


@torch.no_grad()
def gen_pcv(ref):
    import warnings
    print("PCVNet correlation block / CGI normalised volumes")
    out = {}
    for name, c in _cases.PCV_CASES.items():
        f1, f2, coords, sigma = _cases.pcv_inputs(c)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")          # torch.range deprecation inside the reference
            blk = ref.pcv_corr.CorrBlock1D(T(f1), T(f2), sample_num=c["S"], num_levels=c["L"], downsample=c["downsample"])
            look = blk(T(coords), T(sigma)).numpy()
        pyr = [flat(p.numpy()) for p in blk.corr_pyramid]
        assert len(pyr) == c["L"]
        for i, p in enumerate(pyr):
            out["pcv/%s/pyr%d" % (name, i)] = p
        out["pcv/%s/lookup" % name] = look
        tp, factor = to.pcv_pyramid(T(f1), T(f2), c["L"], c["downsample"])
        assert factor == blk.compress_factor
        for i in range(c["L"]):
            pin("torch.pcv_pyramid[%s][%d]" % (name, i), flat(tp[i].numpy()), pyr[i], 0.0)
        pin("torch.pcv_lookup[%s]" % name, to.pcv_lookup(tp, T(coords), T(sigma), c["S"], factor).numpy(), look, 0.0)
        cp = co.pcv_pyramid(pyr[0], c["L"], factor)
        for i in range(c["L"]):
            pin("c.pcv_pool[%s][%d] (from reference level 0)" % (name, i), cp[i], pyr[i], 0.0)
        pin("c.pcv_lookup[%s] (on reference pyramid)" % name, co.pcv_lookup(pyr, coords, sigma, c["S"], factor), look, 0.0)
    for name, c in _cases.NORMCORR_CASES.items():
        a, b = _cases.volume_inputs(c)
        v = ref.cgi_sub.build_gwc_volume_norm(T(a), T(b), c["D"], c["G"]).numpy()
        out["normcorr/%s/gwc_norm" % name] = v
        pin("torch.gwc_volume_norm[%s]" % name, to.gwc_volume_norm(T(a), T(b), c["D"], c["G"]).numpy(), v, 0.0)
        pin("c.gwc_volume_norm[%s]" % name, co.gwc_volume_norm(a, b, c["D"], c["G"]), v, 1e-6)
        if c["G"] == 1:
            v1 = ref.cgi_sub.build_norm_correlation_volume(T(a), T(b), c["D"]).numpy()
            v2 = ref.igev_sub.build_norm_correlation_volume(T(a), T(b), c["D"]).numpy()
            assert np.array_equal(v1, v2)
            out["normcorr/%s/norm_corr" % name] = v1
            pin("torch.norm_correlation_volume[%s]" % name, to.norm_correlation_volume(T(a), T(b), c["D"]).numpy(), v1, 0.0)
            pin("c.norm_correlation_volume[%s]" % name, co.gwc_volume_norm(a, b, c["D"], 1), v1, 1e-6)
    save("pcv_cgi", **out)


@torch.no_grad()
def gen_geo(ref):
    print("IGEV geometry volume")
    out = {}
    for name, c in _cases.GEO_CASES.items():
        m1, m2, geo, disp, coords = _cases.geo_inputs(c)
        g = ref.GeoVolume(T(m1), T(m2), T(geo), num_levels=c["L"], radius=c["r"])
        look = g(T(disp), T(coords)).numpy()
        out["%s/lookup" % name] = look
        out["%s/init0" % name] = flat(g.init_corr_pyramid[0].numpy())
        gp, ip = to.geo_pyramids(T(m1), T(m2), T(geo), c["L"])
        pin("torch.geo_lookup[%s]" % name, to.geo_lookup(gp, ip, T(disp), T(coords), c["r"]).numpy(), look, 0.0)
        # C oracle on the reference's own pyramids isolates the sampler arithmetic
        rgp = [p.numpy().reshape(-1, p.shape[-1]) for p in g.geo_volume_pyramid]
        rip = [flat(p.numpy()) for p in g.init_corr_pyramid]
        pin("c.geo_lookup[%s] (on reference pyramids)" % name,
            co.geo_lookup(rgp, rip, disp, coords, c["C"], c["r"]), look, 0.0)
        cgp, cip = co.geo_pyramids(m1, m2, geo, c["L"])
        for i in range(c["L"]):
            pin("c.geo_pyr[%s][%d]" % (name, i), cgp[i], rgp[i], 0.0)
            pin("c.init_pyr[%s][%d]" % (name, i), cip[i], rip[i], 4e-6 * float(np.abs(rip[0]).max()))
    save("geo", **out)


@torch.no_grad()
def gen_volumes(ref):
    print("cost volumes")
    out = {}
    for name, c in _cases.GWC_CASES.items():
        a, b = _cases.volume_inputs(c)
        v1 = ref.igev_sub.build_gwc_volume(T(a), T(b), c["D"], c["G"]).numpy()
        v2 = ref.gwc_sub.build_gwc_volume(T(a), T(b), c["D"], c["G"]).numpy()
        assert np.array_equal(v1, v2)
        out["gwc/%s" % name] = v1
        pin("torch.gwc_volume[%s]" % name, to.gwc_volume(T(a), T(b), c["D"], c["G"]).numpy(), v1, 0.0)
        pin("c.gwc_volume[%s]" % name, co.gwc_volume(a, b, c["D"], c["G"]), v1, 1e-6)
    for name, c in _cases.CONCAT_CASES.items():
        a, b = _cases.volume_inputs(c)
        vg = ref.gwc_sub.build_concat_volume(T(a), T(b), c["D"]).numpy()
        vi = ref.igev_sub.build_concat_volume(T(a), T(b), c["D"]).numpy()
        out["concat_gwcnet/%s" % name] = vg
        out["concat_igev/%s" % name] = vi
        pin("torch.concat_volume[gwcnet,%s]" % name, to.concat_volume(T(a), T(b), c["D"], True).numpy(), vg, 0.0)
        pin("torch.concat_volume[igev,%s]" % name, to.concat_volume(T(a), T(b), c["D"], False).numpy(), vi, 0.0)
        pin("c.concat_volume[gwcnet,%s]" % name, co.concat_volume(a, b, c["D"], 1), vg, 0.0)
        pin("c.concat_volume[igev,%s]" % name, co.concat_volume(a, b, c["D"], 0), vi, 0.0)
    save("volumes", **out)


def _load(module, seed, prefix=""):
    shapes = _synth.shapes_of(module)
    sd = _synth.torch_state_dict({prefix + k: v for k, v in shapes.items()}, seed)
    module.load_state_dict({k[len(prefix):]: v for k, v in sd.items()}, strict=True)
    module.eval()
    return sd


@torch.no_grad()
def gen_gru(ref):
    print("ConvGRU")
    out = {}
    for name, c in _cases.GRU_CASES.items():
        h, czrq, xs = _cases.gru_inputs(c)
        hd = c["hidden"]
        for flavour, mod in (("raft", ref.update), ("igev", ref.igev_update)):
            gru = mod.ConvGRU(hd, sum(c["inputs"]))
            sd = _load(gru, c["seed"], "g.")
            cz, cr, cq = T(czrq).split(hd, dim=1)
            y = gru(T(h), cz, cr, cq, *[T(x) for x in xs]).numpy()
            if flavour == "raft":
                out["%s/h" % name] = y
                pin("torch.conv_gru[%s]" % name,
                    to.conv_gru(sd, "g", T(h), cz, cr, cq, *[T(x) for x in xs]).numpy(), y, 0.0)
                # gate arithmetic of the C oracle on the reference's conv outputs
                hx = torch.cat([T(h)] + [T(x) for x in xs], 1)
                az, ar = gru.convz(hx), gru.convr(hx)
                z, rh = co.gru_gate_zr(az.numpy(), ar.numpy(), cz.numpy(), cr.numpy(), h)
                aq = gru.convq(torch.cat([T(rh), *[T(x) for x in xs]], 1))
                pin("c.gru_gates[%s]" % name, co.gru_gate_out(aq.numpy(), cq.numpy(), z, h), y, 2e-6)
                pin("c.conv2d_same[%s]" % name,
                    co.conv2d_same(hx.numpy(), gru.convz.weight.numpy(), gru.convz.bias.numpy()), az.numpy(), 2e-5)
            else:
                assert np.array_equal(y, out["%s/h" % name])  # the two ConvGRU copies are the same operator
    save("gru", **out)


@torch.no_grad()
def gen_update(ref):
    print("update blocks")
    out = {}
    for name, c in _cases.UPDATE_CASES.items():
        cfg = _cases.update_cfg(c)
        args = SimpleNamespace(**cfg)
        mod = ref.igev_update if c["igev"] else ref.update
        blk = mod.BasicMultiUpdateBlock(args, hidden_dims=cfg["hidden_dims"])
        sd = _load(blk, c["seed"], "update_block.")
        MANIFEST.setdefault("state_dict_keys", {})[name] = sorted(k[len("update_block."):] for k in sd)
        net, inp, corr, flow = _cases.update_inputs(c)
        n = c["n"]
        tnet = [T(x.copy()) for x in net]
        tinp = [list(T(x).split(128, dim=1)) for x in inp]
        if c["igev"]:
            res = blk(tnet, tinp, T(corr), T(flow), iter16=(n == 3), iter08=(n >= 2))
        else:
            res = blk(tnet, tinp, T(corr), T(flow), iter32=(n == 3), iter16=(n >= 2))
        rnet, rmask, rdelta = res
        for i in range(3):
            out["%s/net%d" % (name, i)] = rnet[i].numpy()
        out["%s/mask" % name] = rmask.numpy()[:, :, ::2, ::2].copy()
        out["%s/mask_stride" % name] = np.int64(2)
        out["%s/delta" % name] = rdelta.numpy()
        onet = [T(x.copy()) for x in net]
        onet, omask, odelta = to.update_block(sd, "update_block", n, onet, tinp, T(corr), T(flow),
                                              it_coarse=(n == 3), it_mid=(n >= 2), igev=c["igev"])
        for i in range(3):
            pin("torch.update_block[%s].net%d" % (name, i), onet[i].numpy(), rnet[i].numpy(), 0.0)
        pin("torch.update_block[%s].mask" % name, omask.numpy(), rmask.numpy(), 0.0)
        pin("torch.update_block[%s].delta" % name, odelta.numpy(), rdelta.numpy(), 0.0)
        # motion encoder alone
        enc = blk.encoder(T(flow), T(corr)).numpy()
        out["%s/motion" % name] = enc[:, ::4].copy()
        pin("torch.motion_encoder[%s]" % name,
            to.motion_encoder(sd, "update_block.encoder", T(flow), T(corr), igev=c["igev"]).numpy(), enc, 0.0)
    save("update", **out)


@torch.no_grad()
def gen_e2e(ref):
    print("RAFT-Stereo end to end (reference RAFTStereo.forward, test_mode)")
    cfg_path = os.path.join(_refimport.REF, "configs", "raft_stereo", "base.json")
    cfg = json.load(open(cfg_path))
    args = SimpleNamespace(mixed_precision=False, **cfg)
    model = ref.RAFTStereo(args)
    shapes = _synth.shapes_of(model)
    sd = _synth.torch_state_dict(shapes, _cases.E2E_WEIGHT_SEED)
    model.load_state_dict(sd, strict=True)
    model.eval()
    MANIFEST["raft_state_dict"] = {k: list(v) for k, v in sorted(shapes.items())}
    out = {}
    for name, c in _cases.E2E_CASES.items():
        i1, i2 = _synth.image_pair(c["seed"], c["B"], c["H"], c["W"], c["shift"])
        flow_lo, flow_up = model(T(i1), T(i2), iters=c["iters"], test_mode=True)
        s = c["stride"]
        out["%s/flow_up" % name] = flow_up.numpy()[:, :, ::s, ::s].copy()
        out["%s/flow_lo" % name] = flow_lo.numpy()[:, :1].copy()
        out["%s/stride" % name] = np.int64(s)
        print("   %s: disparity range %.2f .. %.2f" % (name, float(-flow_up.max()), float(-flow_up.min())))
        o_lo, o_up = to.raft_stereo_forward(sd, cfg, T(i1), T(i2), c["iters"])
        pin("torch.raft_stereo[%s].flow_up" % name, o_up.numpy(), flow_up.numpy(), 1e-5)
        pin("torch.raft_stereo[%s].flow_lo" % name, o_lo.numpy(), flow_lo.numpy(), 1e-5)
        if name == "64x128_it12":
            args_alt = SimpleNamespace(**{**vars(args), "corr_implementation": "alt"})
            m2 = ref.RAFTStereo(args_alt)
            m2.load_state_dict(sd, strict=True)
            m2.eval()
            _, alt_up = m2(T(i1), T(i2), iters=c["iters"], test_mode=True)
            d = float((alt_up - flow_up).abs().max())
            MANIFEST["reference_reg_vs_alt_max_abs"] = d
            out["%s/flow_up_alt" % name] = alt_up.numpy()
            print("   reference reg vs alt: %.3e" % d)
    for name, c in _cases.E2E_SLOWFAST_CASES.items():
        over = dict(slow_fast_gru=True, n_gru_layers=c["n"])
        m3 = ref.RAFTStereo(SimpleNamespace(**{**vars(args), **over}))
        sd3 = _synth.torch_state_dict(_synth.shapes_of(m3), _cases.E2E_WEIGHT_SEED)
        m3.load_state_dict(sd3, strict=True)
        m3.eval()
        i1, i2 = _synth.image_pair(c["seed"], c["B"], c["H"], c["W"], c["shift"])
        flow_lo, flow_up = m3(T(i1), T(i2), iters=c["iters"], test_mode=True)
        out["%s/flow_up" % name] = flow_up.numpy()
        out["%s/flow_lo" % name] = flow_lo.numpy()[:, :1].copy()
        o_lo, o_up = to.raft_stereo_forward(sd3, {**cfg, **over}, T(i1), T(i2), c["iters"])
        pin("torch.raft_stereo[%s].flow_up" % name, o_up.numpy(), flow_up.numpy(), 1e-5)
        pin("torch.raft_stereo[%s].flow_lo" % name, o_lo.numpy(), flow_lo.numpy(), 1e-5)
    save("raft_e2e", **out)


@torch.no_grad()
def gen_backbones(ref):
    print("RAFT-Stereo end to end, shared_backbone / backbone_type='interpolate' (raft_stereo.py:43-54, 97-108)")
    cfg = json.load(open(os.path.join(_refimport.REF, "configs", "raft_stereo", "base.json")))
    out = {}
    for name, c in _cases.E2E_BACKBONE_CASES.items():
        m = ref.RAFTStereo(SimpleNamespace(mixed_precision=False, **{**cfg, **c["over"]}))
        sd = _synth.torch_state_dict(_synth.shapes_of(m), _cases.E2E_WEIGHT_SEED)
        m.load_state_dict(sd, strict=True)
        m.eval()
        i1, i2 = _synth.image_pair(c["seed"], c["B"], c["H"], c["W"], c["shift"])
        flow_lo, flow_up = m(T(i1), T(i2), iters=c["iters"], test_mode=True)
        out["%s/flow_up" % name] = flow_up.numpy()
        out["%s/flow_lo" % name] = flow_lo.numpy()[:, :1].copy()
        MANIFEST["raft_state_dict[%s]" % name] = {k: list(v) for k, v in sorted(_synth.shapes_of(m).items())}
        print("   %s: disparity range %.2f .. %.2f" % (name, float(-flow_up.max()), float(-flow_up.min())))
    save("raft_backbones", **out)


@torch.no_grad()
def gen_igev_loop(ref):
    print("IGEV GRU loop from match features onward")
    out = {}
    for name, c in _cases.IGEV_LOOP_CASES.items():
        s = c["seed"]
        cfg = _cases.igev_loop_cfg(c)
        n, slow_fast = cfg["n_gru_layers"], cfg["slow_fast_gru"]
        args = SimpleNamespace(**cfg)
        blk = ref.igev_update.BasicMultiUpdateBlock(args, hidden_dims=cfg["hidden_dims"])
        sd = _load(blk, s, "update_block.")
        m1, m2, geo, disp0, coords, net, inp = _cases.igev_loop_inputs(c)
        tnet = [T(x.copy()) for x in net]
        tinp = [list(T(x).split(128, dim=1)) for x in inp]
        # the loop of igev_stereo.py:192-210, written against the reference's own classes
        geo_fn = ref.GeoVolume(T(m1), T(m2), T(geo), radius=4, num_levels=2)
        d = T(disp0)
        for _ in range(c["iters"]):
            feat = geo_fn(d, T(coords))
            if n == 3 and slow_fast:
                tnet = blk(tnet, tinp, iter16=True, iter08=False, iter04=False, update=False)
            if n >= 2 and slow_fast:
                tnet = blk(tnet, tinp, iter16=(n == 3), iter08=True, iter04=False, update=False)
            tnet, mask, delta = blk(tnet, tinp, feat, d, iter16=(n == 3), iter08=(n >= 2))
            d = d + delta
        st = c["stride"]
        out["%s/disp" % name] = d.numpy()
        out["%s/mask" % name] = mask.numpy()[:, :, ::st, ::st].copy()
        out["%s/mask_stride" % name] = np.int64(st)
        print("   %s: disparity range %.2f .. %.2f" % (name, float(d.min()), float(d.max())))
        od, om = to.igev_iterations(sd, cfg, T(m1), T(m2), T(geo), T(disp0),
                                    [T(x.copy()) for x in net], tinp, c["iters"])
        pin("torch.igev_iterations[%s].disp" % name, od.numpy(), d.numpy(), 0.0)
        pin("torch.igev_iterations[%s].mask" % name, om.numpy(), mask.numpy(), 0.0)
    save("igev_loop", **out)


@torch.no_grad()
def gen_gwcnet(ref):
    """GwcNet end to end (BASELINE configs[4]): the reference GWCNet in eval mode, test_mode=True."""
    import contextlib
    import io
    print("GwcNet end to end (reference GWCNet.forward, eval, test_mode)")
    args = SimpleNamespace(maxdisp=192, use_concat_volume=True, mixed_precision=False)
    with contextlib.redirect_stdout(io.StringIO()):
        model = ref.GWCNet(args)
    shapes = _synth.shapes_of(model)
    sd = _synth.torch_state_dict(shapes, _cases.GWCNET_WEIGHT_SEED)
    model.load_state_dict(sd, strict=True)
    model.eval()
    MANIFEST["gwcnet_state_dict"] = {k: list(v) for k, v in sorted(shapes.items())}
    out = {}
    for name, c in _cases.GWCNET_CASES.items():
        i1, i2 = _synth.image_pair(c["seed"], c["B"], c["H"], c["W"], c["shift"])
        _, disp = model(T(i1), T(i2), test_mode=True)
        s = c["stride"]
        out["%s/disp" % name] = disp.numpy()[:, :, ::s, ::s].copy()
        out["%s/stride" % name] = np.int64(s)
        print("   %s: disparity range %.2f .. %.2f" % (name, float(-disp.max()), float(-disp.min())))
    save("gwcnet", **out)


@torch.no_grad()
def gen_eval(ref):
    """The evaluator's chain on a size that needs padding: reference InputPadder(divis_by=32) -> reference
    RAFTStereo.forward(test_mode) -> unpad (tools/evaluate_stereo.py:124-134)."""
    print("evaluation chain: pad /32 -> RAFT-Stereo -> unpad")
    cfg = json.load(open(os.path.join(_refimport.REF, "configs", "raft_stereo", "base.json")))
    model = ref.RAFTStereo(SimpleNamespace(mixed_precision=False, **cfg))
    sd = _synth.torch_state_dict(_synth.shapes_of(model), _cases.E2E_WEIGHT_SEED)
    model.load_state_dict(sd, strict=True)
    model.eval()
    out = {}
    for name, c in _cases.EVAL_CASES.items():
        i1, i2 = _synth.image_pair(c["seed"], 1, c["H"], c["W"], c["shift"])
        i1, i2 = np.floor(i1), np.floor(i2).clip(0, 255)          # what an 8-bit image file holds
        padder = ref.utils.InputPadder(i1.shape, divis_by=32)
        p1, p2 = padder.pad(T(i1), T(i2))
        _, flow_pr = model(p1, p2, iters=c["iters"], test_mode=True)
        out["%s/flow" % name] = padder.unpad(flow_pr).numpy()
        out["%s/padded_shape" % name] = np.array(p1.shape[-2:], np.int64)
    save("eval", **out)


def gen_sampler(ref):
    print("sampler bit-exactness (C oracle vs reference bilinear_sampler)")
    g = _synth.rng(99, "sampler")
    W, N = 39, 20000
    rows = g.standard_normal((N, W)).astype(np.float32)
    x = g.uniform(-6, W + 5, N).astype(np.float32)
    x[::50] = np.round(x[::50])
    img = T(rows).view(N, 1, 1, W)
    c = torch.stack([T(x), torch.zeros(N)], -1).view(N, 1, 1, 2)
    want = ref.utils.bilinear_sampler(img, c).view(N).numpy()
    got = np.array([co.bilinear_1d(rows[i], float(x[i])) for i in range(N)], np.float32)
    pin("c.bilinear_1d (20k samples)", got, want, 0.0)
    save("sampler", x=x[:2000], out=want[:2000])


def main():
    if not _refimport.available():
        raise SystemExit("reference tree not found at %s -- fixtures can only be generated in the "
                         "build container" % _refimport.REF)
    co.build()
    ref = _refimport.load()
    only = set(sys.argv[1:])
    gens = [("sampler", gen_sampler), ("corr", gen_corr), ("geo", gen_geo), ("volumes", gen_volumes), ("pcv", gen_pcv), ("corr_bwd", gen_corr_bwd), ("geo_bwd", gen_geo_bwd), ("upsample", gen_upsample), ("files", gen_files),
            ("gru", gen_gru), ("update", gen_update), ("igev_loop", gen_igev_loop), ("e2e", gen_e2e), ("backbones", gen_backbones), ("gwcnet", gen_gwcnet), ("eval", gen_eval)]
    for name, fn in gens:
        if not only or name in only:
            fn(ref)
    path = os.path.join(HERE, "MANIFEST.json")
    if only and os.path.exists(path):                 # partial run: merge into the recorded pins
        old = json.load(open(path))
        old["pins"].update(MANIFEST["pins"])
        for k, v in old.items():                      # keep everything the skipped generators recorded
            if k != "pins" and k not in MANIFEST:
                MANIFEST[k] = v
        MANIFEST["pins"] = old["pins"]
    with open(path, "w") as f:
        json.dump(MANIFEST, f, indent=1, sort_keys=True)
    print("wrote MANIFEST.json (%d pins)" % len(MANIFEST["pins"]))


if __name__ == "__main__":
    main()
