"""MFMA group-wise correlation (gwc_mfma.hip) vs the bit-exact VALU kernel and an fp64 reference; timing of both."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from c8_check import gtime
from dkt_stereo_amd import submodule as sm
torch.manual_seed(0)
with torch.no_grad():
    for name, (B, C, H, W, G) in {"IGEV 96ch G=8": (1, 96, 184, 312, 8), "GwcNet 320ch G=40": (1, 320, 136, 240, 40),
                                  "small 32ch G=4 B=2": (2, 32, 37, 100, 4), "cpg16": (1, 64, 20, 68, 4)}.items():
        a, b = torch.randn(B, C, H, W, device="cuda:0"), torch.randn(B, C, H, W, device="cuda:0")
        sm.GWC_MODE = "exact"; ve = sm.build_gwc_volume(a, b, 48, G)
        sm.GWC_MODE = "mfma"; vm = sm.build_gwc_volume(a, b, 48, G)
        # fp64 reference
        ref = torch.zeros(B, G, 48, H, W, device="cuda:0", dtype=torch.float64)
        ad, bd = a.double().view(B, G, C // G, H, W), b.double().view(B, G, C // G, H, W)
        for d in range(min(48, W)):
            ref[:, :, d, :, d:] = (ad[..., d:] * bd[..., :W - d]).mean(2)
        sc = float(ref.abs().max())
        print("%-22s mfma vs fp64 %.2e  exact vs fp64 %.2e  mfma vs exact %.2e (scale %.2f)" % (
            name, float((vm - ref).abs().max()) / sc, float((ve - ref).abs().max()) / sc, float((vm - ve).abs().max()) / sc, sc), end="")
        if H > 100:
            sm.GWC_MODE = "exact"; te = gtime(lambda: sm.build_gwc_volume(a, b, 48, G), 5, 6)
            sm.GWC_MODE = "mfma"; tm = gtime(lambda: sm.build_gwc_volume(a, b, 48, G), 5, 6)
            out_mb = B * G * 48 * H * W * 4 / 1e6
            in_mb = 2 * B * C * H * W * 4 / 1e6
            flops = 2.0 * B * C * 48 * H * W
            issued = B * G * H * ((W + 63) // 64) * 16 * (C // G // 4) * (2 * 16 * 16 * 4)
            print("  | exact %.1f us, mfma %.1f us = %.2f TB/s of output (%.2f incl. reads), MFMA issued %.1f TF/s = %.1f %% of the 157 TF fp32 matrix peak"
                  % (te, tm, out_mb / tm, (out_mb + in_mb) / tm, issued / tm / 1e6, issued / tm / 1e6 / 157.3 * 100))
        else:
            print()
