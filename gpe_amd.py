"""Import alias: the package directory is named `garment-pattern-estimation_amd` (not a valid identifier), so
`import gpe_amd` loads it under this name:  `from gpe_amd import nets, net_blocks, ops`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'garment-pattern-estimation_amd')
_spec = importlib.util.spec_from_file_location('gpe_amd', os.path.join(_dir, '__init__.py'),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['gpe_amd'] = _mod
_spec.loader.exec_module(_mod)
