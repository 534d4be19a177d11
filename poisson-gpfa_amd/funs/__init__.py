"""Drop-in for the `funs` package of mackelab/poisson-gpfa, EM hot path only, running on MI355X.

    import funs.util as util
    import funs.engine as engine

works as with the reference when the directory that contains this package
(`poisson-gpfa_amd/`) is on sys.path.  Compute goes through libpgpfa_hip.so (HIP, gfx950);
there is no CPU fallback.
"""
from . import util, inference, learning, engine, mcmc  # noqa: F401

__all__ = ['util', 'inference', 'learning', 'engine', 'mcmc']
