"""Drop-in module `impl.utils`: same import path and names as the reference's impl/utils.py, backed by
glass_amd.utils (MI355X HIP path)."""
import sys as _sys

from glass_amd import utils as _impl

_sys.modules[__name__] = _impl
