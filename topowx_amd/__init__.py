"""topowx_amd: MI355X-native moving-window regression-kriging / GWR interpolator.

Drop-in for the ``twx.interp`` hot path of jaredwo/topowx.
"""
__version__ = "0.1.0"
