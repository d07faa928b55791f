"""``twx.interp`` = ``topowx_amd.interp`` (twx/interp/__init__.py of the reference star-imports its four modules)."""
from topowx_amd.interp import *  # noqa: F401,F403
from topowx_amd.interp import __all__  # noqa: F401
