from .seed import seed_everything  # noqa: F401
from .tracker import Tracker  # noqa: F401
