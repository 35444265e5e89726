"""3-D math helpers with the reference's flat namespace (`three.transform_coords`, `three.quaternion.qmul`, ...)."""
import importlib as _importlib

# sub-modules reachable as attributes, and the function sets that the reference re-exports at package level
quaternion, orientation, utils = (_importlib.import_module(f'{__name__}.{m}') for m in ('quaternion', 'orientation', 'utils'))
for _name in ('core', 'rigid', 'batchview'):
    _mod = _importlib.import_module(f'{__name__}.{_name}')
    globals().update({k: v for k, v in vars(_mod).items() if not k.startswith('_')})
del _importlib, _name, _mod
