"""Import shim: the package directory is named `gradient-based-path-tracing_amd` (hyphens), which Python
cannot import by name; `import gdpt_amd` loads it from that directory under this alias."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gradient-based-path-tracing_amd")
_spec = importlib.util.spec_from_file_location("gdpt_amd", os.path.join(_pkg_dir, "__init__.py"),
                                               submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["gdpt_amd"] = _mod
_spec.loader.exec_module(_mod)
