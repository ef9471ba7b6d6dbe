"""bhnerf_amd: MI355X-native engine for the bhnerf hot path.

Module names mirror the reference package (bhnerf.network / emission / kgeo / optimization /
utils / constants) for the functions on the hot path.
"""
from . import constants, units, utils  # noqa: F401
from . import checkpoints, emission, geodesics, kgeo, network, observation, optimization, alma  # noqa: F401

__version__ = '0.1.0'
