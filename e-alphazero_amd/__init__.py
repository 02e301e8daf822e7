"""e-alphazero_amd: MI355X-native self-play engine behind the call surface of emcts/e-alphazero's hot path.

Import as `e_alphazero_amd` (the root-level e_alphazero_amd.py aliases this hyphenated directory).
The compute path is libeaz.so (hand-written HIP for gfx950, C ABI in include/eaz.h); this package is the host-side
mirror of the reference's Python interface (pgx.Env, forward.apply, emctx policy, selfplay).
"""
from . import params  # noqa: F401  (pure numpy, importable without a GPU)
from ._lib import EazError, LIB_PATH, SIGNATURES, load as load_library  # noqa: F401

__all__ = ["params", "EazError", "LIB_PATH", "SIGNATURES", "load_library"]


def __getattr__(name):
    # torch-dependent modules are imported lazily so `import e_alphazero_amd` stays cheap
    if name in ("engine", "api"):
        import importlib

        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
