"""Drop-in module `impl.models`: same import path and names as the reference's impl/models.py, backed by
glass_amd.models (MI355X HIP path)."""
import sys as _sys

from glass_amd import models as _impl

_sys.modules[__name__] = _impl
