"""datr_amd -- MI355X-native implementation of DATR's data-parallel training hot path.

Importing the package loads libdatr_hip.so (datr_amd/_native.py); there is no fallback.
"""
from . import _native  # noqa: F401  (fails loudly when the HIP library is missing)

__version__ = "0.1.0"
