"""lorikeet_amd -- MI355X-native PairHMM read x haplotype likelihood engine (drop-in for the
PairHMM path of rhysnewell/Lorikeet).  See DESIGN.md / INTEGRATION.md.

Only what the path needs lives here: `csrc/` (gfx950 kernels + the C ABI of include/phmm.h) and
the host-side mirror of the reference interface.  Importing the package does not need a GPU;
creating an engine does (there is no CPU fallback).
"""
from .batch import Read, RegionBatch  # noqa: F401
from .engine import HipPairHMMEngine, PhmmError  # noqa: F401
