"""MI355X-native volumetric photon mapping: Python-side plumbing (ctypes) over the C ABI
declared in include/pvol.h.  The product is csrc/ (HIP + C-ABI); this package only loads it."""
from . import abi, blob  # noqa: F401
