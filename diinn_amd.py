"""Import alias: ``import diinn_amd`` loads the package that lives in the
(hyphenated, hence not directly importable) directory
``dual-interactive-implicit-neural-network_amd/``."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_DIR = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)),
                     "dual-interactive-implicit-neural-network_amd")
_spec = _ilu.spec_from_file_location("diinn_amd", _os.path.join(_DIR, "__init__.py"),
                                     submodule_search_locations=[_DIR])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["diinn_amd"] = _mod
_spec.loader.exec_module(_mod)
