"""Import shim for the upstream reference (TEST INFRASTRUCTURE ONLY).

Makes ``/root/reference/latentfusion`` importable inside THIS build container so that
``oracle/make_golden.py`` can generate golden vectors from the real PyTorch reference
(SURVEY.md section 8c / Appendix B).  Nothing here travels to the GPU box in a usable form:
the reference tree does not exist there and ``load_reference()`` raises.

The shim never writes into ``/root/reference`` (bytecode writing is disabled) and never
copies reference sources: it only installs empty stand-ins for third-party modules the
image lacks (structlog, toml, cv2, ...), which the reference imports at module scope but the
reconstruct-and-render path never calls.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types
from unittest import mock

REFERENCE_ROOT = '/root/reference'

# Import-only dependencies of the reference that are absent from this image.
_ABSENT = ('imageio', 'cv2', 'plyfile', 'torchvision', 'skimage', 'pyrender', 'trimesh',
           'torchnet', 'seaborn', 'av', 'tensorboard', 'tensorboardX', 'OpenGL', 'pyglet')


class _NullLogger:
    def _noop(self, *a, **k):
        return None
    info = warning = error = debug = exception = critical = _noop

    def bind(self, **k):
        return self


class _MockLoader(importlib.abc.Loader):
    def create_module(self, spec):
        m = mock.MagicMock()
        m.__path__ = []
        m.__spec__ = spec
        m.__name__ = spec.name
        return m

    def exec_module(self, module):
        pass


class _MockFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, name, path, target=None):
        if name.split('.')[0] in _ABSENT:
            return importlib.machinery.ModuleSpec(name, _MockLoader(), is_package=True)
        return None


_loaded = None


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'latentfusion'))


def load_reference():
    """Returns the imported ``latentfusion`` reference package (cached)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not reference_available():
        raise RuntimeError('reference tree not present (expected only in the build container)')
    sys.dont_write_bytecode = True

    # structlog: used at import time for logger construction/configuration only.
    if 'structlog' not in sys.modules:
        sl = types.ModuleType('structlog')
        sl.get_logger = lambda *a, **k: _NullLogger()
        sl.configure = lambda **k: None
        for sub in ('stdlib', 'processors', 'dev'):
            setattr(sl, sub, mock.MagicMock())
        sys.modules['structlog'] = sl
    # IPython: is_notebook() probes get_ipython.
    if 'IPython' not in sys.modules:
        ip = types.ModuleType('IPython')
        ip.get_ipython = lambda: None
        sys.modules['IPython'] = ip
    # toml -> tomli.
    if 'toml' not in sys.modules:
        import tomli
        tm = types.ModuleType('toml')

        def _load(path):
            with open(path, 'rb') as f:
                return tomli.load(f)
        tm.load = _load
        tm.loads = tomli.loads
        sys.modules['toml'] = tm
    sys.meta_path.insert(0, _MockFinder())

    # torch >= 2.x removed ReduceLROnPlateau(verbose=...); the reference still passes it.
    from torch.optim import lr_scheduler
    base = lr_scheduler.ReduceLROnPlateau
    if not getattr(base, '_lf_compat', False):
        class _Plateau(base):
            _lf_compat = True

            def __init__(self, *a, verbose=None, **k):
                super().__init__(*a, **k)
        lr_scheduler.ReduceLROnPlateau = _Plateau

    # torch >= 2.x: Sampler.__init__ takes no data_source any more; the reference's samplers still pass one
    # (torchutils.py:181,199 -- used by Observation.from_dataset through IndexedDataLoader).
    from torch.utils.data import Sampler
    if not getattr(Sampler, '_lf_compat', False):
        Sampler.__init__ = lambda self, *a, **k: None
        Sampler._lf_compat = True

    import warnings
    warnings.filterwarnings('ignore')
    sys.path.insert(0, REFERENCE_ROOT)
    import latentfusion  # noqa
    import latentfusion.recon.inference  # noqa
    import latentfusion.pose.estimation  # noqa
    _loaded = latentfusion
    return latentfusion
