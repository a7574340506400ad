"""dkt_stereo_amd -- MI355X (gfx950) implementation of DKT-Stereo's stereo
inference hot path: correlation volume build, per-iteration lookup and the
ConvGRU update operator, behind the reference's Python class signatures.

    dkt_stereo_amd.corr        CorrBlock1D & variants        (core/corr.py)
    dkt_stereo_amd.geometry    Combined_Geo_Encoding_Volume  (meta_arch/igev_stereo/geometry.py)
    dkt_stereo_amd.submodule   build_gwc_volume / build_concat_volume
    dkt_stereo_amd.update      ConvGRU, BasicMotionEncoder, BasicMultiUpdateBlock (+IGEV)
    dkt_stereo_amd.raft_stereo RAFTStereo harness (same state_dict keys as the reference)

All compute goes through libdktstereo.so (include/dktstereo.h); see DESIGN.md.
"""
__version__ = "0.1.0"
