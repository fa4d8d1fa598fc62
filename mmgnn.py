"""Import alias: ``import mmgnn`` loads the package that lives in ``multi-modal-gnn_amd/``
(the directory name the project layout prescribes is not a valid Python identifier)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "multi-modal-gnn_amd")
_spec = importlib.util.spec_from_file_location("mmgnn", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mmgnn"] = _mod
_spec.loader.exec_module(_mod)
