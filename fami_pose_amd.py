"""Import shim: the package directory is `fami-pose_amd/` (hyphen, as the repo
layout names it), which Python cannot import by name.  `import fami_pose_amd`
loads that directory as the package `fami_pose_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'fami-pose_amd')
_spec = importlib.util.spec_from_file_location(
    'fami_pose_amd', os.path.join(_dir, '__init__.py'), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['fami_pose_amd'] = _mod
_spec.loader.exec_module(_mod)
