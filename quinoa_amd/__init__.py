"""quinoa_amd -- MI355X-native DG compressible-flow path for Quinoa/Inciter.

The product is quinoa_amd/lib/libqdg.so (C ABI: include/qdg.h; HIP kernels in
quinoa_amd/csrc/).  The Python modules here are the ctypes binding and the
host-side chunk assembly used by tests/ and bench.py.
"""
from . import capi  # noqa: F401
