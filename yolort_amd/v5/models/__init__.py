from .common import *  # noqa: F401,F403
