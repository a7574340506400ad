"""Import shim for the upstream reference -- used ONLY by make_golden.py, in the
build container where /root/reference is mounted.  Nothing here (or anything
it imports from the reference) travels to the GPU box; tests read only the
.npz files this directory holds.

The reference's package __init__ files are broken (SURVEY.md section 4:
meta_arch/gwcnet/__init__.py imports a name that does not exist; timm and
opt_einsum are not installed), so the packages are registered bare and the
leaf modules imported directly.
"""
import os
import sys
import types

REF = os.environ.get("DKT_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF, "core"))


def setup():
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF)
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for name, attrs in (("timm", {}), ("opt_einsum", {"contract": None})):
        if name not in sys.modules:
            m = types.ModuleType(name)
            for k, v in attrs.items():
                setattr(m, k, v)
            sys.modules[name] = m
    for pkg in ("meta_arch", "meta_arch.raft_stereo", "meta_arch.igev_stereo", "meta_arch.gwcnet",
                "meta_arch.pcvnet", "meta_arch.pcvnet.utils", "meta_arch.cgi"):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = [os.path.join(REF, *pkg.split("."))]
            sys.modules[pkg] = m


def load_frame_utils():
    """core/utils/frame_utils.py with inert stand-ins for its cv2 / imageio imports (only the
    pure numpy / PIL readers are exercised)."""
    setup()
    for name in ("cv2", "imageio"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            if name == "cv2":
                m.setNumThreads = lambda n: None
                m.ocl = types.SimpleNamespace(setUseOpenCL=lambda b: None)
            sys.modules[name] = m
    import core.utils.frame_utils as fu
    return fu


def load():
    """Returns a namespace with the reference's hot-path callables."""
    setup()
    ns = types.SimpleNamespace()
    import core.corr as ccorr
    import core.update as cupdate
    import core.utils.utils as cutils
    from meta_arch.raft_stereo.raft_stereo import RAFTStereo
    from meta_arch.igev_stereo.geometry import Combined_Geo_Encoding_Volume
    import meta_arch.igev_stereo.update as iupdate
    import meta_arch.igev_stereo.submodule as isub
    import meta_arch.gwcnet.submodules as gsub
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import meta_arch.pcvnet.corr as pcorr
        import meta_arch.cgi.submodule as csub
    ns.pcv_corr = pcorr
    ns.cgi_sub = csub
    ns.corr = ccorr
    ns.update = cupdate
    ns.utils = cutils
    ns.RAFTStereo = RAFTStereo
    ns.GeoVolume = Combined_Geo_Encoding_Volume
    ns.igev_update = iupdate
    ns.igev_sub = isub
    ns.gwc_sub = gsub
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        from meta_arch.gwcnet.gwc_main import GWCNet
    ns.GWCNet = GWCNet
    return ns
