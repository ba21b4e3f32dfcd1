"""Import helper: the package directory `automationlabsmodelpredictivecontrol.jl_amd/` has a dot in its
name, so it is loaded by path and registered in sys.modules as `almpc_amd`."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(_ROOT, "automationlabsmodelpredictivecontrol.jl_amd")


def load_package():
    if "almpc_amd" in sys.modules:
        return sys.modules["almpc_amd"]
    spec = importlib.util.spec_from_file_location("almpc_amd", os.path.join(PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["almpc_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
