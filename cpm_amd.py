"""Import shim: the package directory is named after the reference (with hyphens), which
``import`` cannot spell.  ``import cpm_amd`` gives the package object."""
import importlib
import sys

_NAME = "correlated-photon-mapping-for-interactive-global-illumination-of-time-varying-volumetric-data_amd"
_pkg = importlib.import_module(_NAME)
sys.modules[__name__] = _pkg
