"""tendrils_amd - MI355X-native drop-in for the GPGPU particle path of keeffEoghan/tendrils.

The product is the gfx950 shared library behind include/tendrils_hip.h
(tendrils_amd/csrc -> tendrils_amd/lib/libtendrils_hip.so).  This package is the
Python host mirror of the reference's JS interface for that path
(`Particles`, `Tendrils`, `Timer`, spawners, `OpticalFlow`), used by the tests
and bench.py; the Node host (tendrils_amd/js) binds the same C ABI through N-API.
"""
from ._capi import (INERT, TH_MODE_EXACT, TH_MODE_FAST, TH_SOURCE_FLOW, TH_STATE_F16, TH_STATE_F32,
                    TH_TARGET_RING, TH_TARGET_TARGETS, TendrilsHipError)
from .particles import Particles, defaults as particles_defaults
from .tendrils import Tendrils, defaults, gl_settings
from .timer import Timer

__all__ = ["Particles", "Tendrils", "Timer", "defaults", "particles_defaults", "gl_settings",
           "TendrilsHipError", "INERT", "TH_MODE_EXACT", "TH_MODE_FAST", "TH_STATE_F32", "TH_STATE_F16", "TH_TARGET_RING",
           "TH_TARGET_TARGETS", "TH_SOURCE_FLOW"]
