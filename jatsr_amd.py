"""Import alias: `import jatsr_amd` loads the package directory
`jatsr-just-audio-transformer-super-solution_amd/` (whose name is not a Python identifier)."""
import importlib.util as _u
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "jatsr-just-audio-transformer-super-solution_amd")
_spec = _u.spec_from_file_location("jatsr_amd", _os.path.join(_dir, "__init__.py"),
                                   submodule_search_locations=[_dir])
_mod = _u.module_from_spec(_spec)
_sys.modules["jatsr_amd"] = _mod
_spec.loader.exec_module(_mod)
