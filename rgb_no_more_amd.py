"""Import shim: the package directory is `rgb-no-more_amd/` (not a valid Python identifier), so
`import rgb_no_more_amd` loads that directory as a package under this name."""
import importlib.util as _u
import os as _os
import sys as _sys

_d = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "rgb-no-more_amd")
_spec = _u.spec_from_file_location(__name__, _os.path.join(_d, "__init__.py"), submodule_search_locations=[_d])
_mod = _u.module_from_spec(_spec)
_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
