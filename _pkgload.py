"""Import the package directory `mitsuba-renderer_amd/` (not a valid Python identifier)
under the module name `mitsuba_renderer_amd`."""
import importlib.util
import os
import sys

NAME = "mitsuba_renderer_amd"


def load():
    if NAME in sys.modules:
        return sys.modules[NAME]
    root = os.path.dirname(os.path.abspath(__file__))
    pkgdir = os.path.join(root, "mitsuba-renderer_amd")
    spec = importlib.util.spec_from_file_location(NAME, os.path.join(pkgdir, "__init__.py"),
                                                  submodule_search_locations=[pkgdir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[NAME] = mod
    spec.loader.exec_module(mod)
    return mod
