"""Drop-in module `impl.train`: same import path and names as the reference's impl/train.py, backed by
glass_amd.train (MI355X HIP path)."""
import sys as _sys

from glass_amd import train as _impl

_sys.modules[__name__] = _impl
