"""Drop-in module `impl.metrics`: same import path and names as the reference's impl/metrics.py, backed by
glass_amd.metrics (MI355X HIP path)."""
import sys as _sys

from glass_amd import metrics as _impl

_sys.modules[__name__] = _impl
