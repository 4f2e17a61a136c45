"""Drop-in module `impl.config`: same import path and names as the reference's impl/config.py, backed by
glass_amd.config (MI355X HIP path)."""
import sys as _sys

from glass_amd import config as _impl

_sys.modules[__name__] = _impl
