from .config import *  # noqa: F401,F403
from .config import Config
