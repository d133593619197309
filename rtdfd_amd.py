"""Importable alias for the package directory ``real-time-video-deepfake-detection_amd``
(its name is not a Python identifier).  ``import rtdfd_amd`` returns that package;
``rtdfd_amd.model`` etc. are its submodules (imported once, under the real name)."""
import importlib
import sys

_REAL = "real-time-video-deepfake-detection_amd"
_pkg = importlib.import_module(_REAL)


class _Alias(type(sys)):
    def __getattr__(self, name):
        try:
            return getattr(_pkg, name)
        except AttributeError:
            try:
                return importlib.import_module(f"{_REAL}.{name}")
            except ModuleNotFoundError as e:
                raise AttributeError(name) from e


_m = _Alias(__name__)
_m.__dict__.update({"__doc__": __doc__, "__file__": __file__, "REAL_NAME": _REAL, "package": _pkg})
sys.modules[__name__] = _m
