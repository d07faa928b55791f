"""``twx`` -- the reference's import names on topowx_amd.

A py3 translation of the reference's ``scripts/step21..27`` keeps its import lines (``from twx.interp import Tiler,
TileWriter, StationDataWrkChk, PtInterpTair``; ``from twx.db import StationSerialDataDb, STN_ID, ...``; ``from twx.utils
import StatusCheck, Unbuffered, TwxConfig``): these packages only re-export ``topowx_amd`` -- no code lives here.
"""
from topowx_amd import __version__  # noqa: F401
