"""Import alias: `import gsbp_amd` loads the package directory `3dgs-gradient-backprojection_amd/`
(whose mandated name is not a valid Python identifier)."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "3dgs-gradient-backprojection_amd")
_spec = _ilu.spec_from_file_location("gsbp_amd", _os.path.join(_dir, "__init__.py"),
                                     submodule_search_locations=[_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["gsbp_amd"] = _mod
_spec.loader.exec_module(_mod)
