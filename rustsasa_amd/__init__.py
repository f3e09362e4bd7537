"""MI355X-native Shrake-Rupley SASA engine behind RustSASA's hot-path API.

The compute lives in rustsasa_amd/lib/librustsasa_amd.so (HIP, gfx950 only,
built from rustsasa_amd/csrc).  Importing this package does not load the
library; the first call does, and raises if it is missing or no GPU is usable.
"""
from .engine import ATOM_DTYPE, Context, RsasaError, device_count, make_atoms, sphere_points

__all__ = ["ATOM_DTYPE", "Context", "RsasaError", "device_count", "make_atoms", "sphere_points"]
