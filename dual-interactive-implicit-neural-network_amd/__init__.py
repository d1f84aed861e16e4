"""MI355X-native DIINN implicit-decoder path (gfx950 HIP kernels behind the
reference's ``ImplicitDecoder`` / ``DIINN`` / ``SRLitModule`` interface).

Heavy imports (torch, the HIP shared library) happen in the submodules, on use.
"""
__version__ = "0.1.0"
