"""ER-NeRF inference path on MI355X: the four extension modules the reference's wrappers import
(`_raymarching_face`, `_gridencoder`, `_shencoder`, `_freqencoder`), implemented over libmerefusion_hip.so."""
