"""lf_oracle -- CPU restatement of the reference reconstruct-and-render path.

TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; the product (latentfusion_amd/) never does.

Parity status: PINNED.  Every function here is checked by tests/test_oracle_golden.py against
golden vectors produced by the real reference (oracle/make_golden.py, which imports
/root/reference in the build container; fixtures G0-G10 under tests/golden/).
"""
from . import camera, ibr, nets, pose, quat  # noqa: F401
from .camera import Cam  # noqa: F401


def cam_from_dict(d):
    """Rebuilds a Cam from the fixture camera record."""
    return Cam(d['K'].clone(), d['log_q'].clone(), d['t'].clone(), viewport=d['viewport'].clone(),
               z_span=d['z_span'], width=d['width'], height=d['height'])
