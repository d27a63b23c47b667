"""shpair — host-side Python mirror of the `pair_style sh` drop-in boundary.

The product is the C-ABI library ``libshpair.so`` (include/shpair.h) built from
``../csrc``; this package is the thin ctypes binding plus the synthetic-bed and
domain-decomposition helpers the tests and bench.py drive it with.  Nothing in
here computes forces on the CPU: if the HIP library is missing or no GPU is
present, construction fails loudly.
"""
from .capi import ShPair, ShPairError, load_library, library_path  # noqa: F401
from . import shapes, bed  # noqa: F401
