"""vectorgraphlibrary_amd -- MI355X (gfx950) backend for VectorGraphLibrary's frontier-driven
advance / compute / reduce / generate_new_frontier hot path.

The product is libvgl_hip.so (hand-written HIP kernels behind the C ABI of include/vgl_hip.h) plus the
C++ drop-in operator class in vectorgraphlibrary_amd/hip/.  This Python package is the thin host harness
used by tests/ and bench.py; it fails loudly when the HIP library or a GPU is missing.
"""
from .lib import VglHipError, LIB_PATH, EXPORTED_SYMBOLS, load  # noqa: F401

__all__ = ["VglHipError", "LIB_PATH", "EXPORTED_SYMBOLS", "load"]
