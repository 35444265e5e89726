from .core import *  # noqa: F401,F403
from . import quaternion  # noqa: F401
from .rigid import *  # noqa: F401,F403
from .batchview import *  # noqa: F401,F403
from . import orientation  # noqa: F401
from . import utils  # noqa: F401
