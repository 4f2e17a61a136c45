"""Drop-in module `impl.SubGDataset`: same import path and names as the reference's impl/SubGDataset.py, backed by
glass_amd.SubGDataset (MI355X HIP path)."""
import sys as _sys

from glass_amd import SubGDataset as _impl

_sys.modules[__name__] = _impl
