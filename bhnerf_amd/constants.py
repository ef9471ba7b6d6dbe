"""Physical constants used on the hot path (reference: bhnerf/constants.py).

The reference evaluates ``G*M/c**3`` with astropy (constants.py:13,17).  astropy is not a
dependency here: the single scalar the hot path needs -- GM/c^3 for Sgr A* in hours
(emission.py:183-185) -- is stored as the value the reference's own expression produces
(tests/golden/g0_constants.npz pins it).
"""
import numpy as np

# constants.py:7-10 -- ISCO radii (pure NumPy in the reference as well)
z1 = lambda a: 1 + (1 - a ** 2) ** (1 / 3) * ((1 + a) ** (1 / 3) + (1 - a) ** (1 / 3))
z2 = lambda a: np.sqrt(3 * a ** 2 + z1(a) ** 2)
isco_pro = lambda a: (3 + z2(a) - np.sqrt((3 - z1(a)) * (3 + z1(a) + 2 * z2(a))))
isco_retro = lambda a: (3 + z2(a) + np.sqrt((3 - z1(a)) * (3 + z1(a) + 2 * z2(a))))

sgra_mass_msun = 4.154e6                       # constants.py:17
GM_c3_hr = 0.0056834692768060625               # GM_c3(sgra_mass).to('hr').value
_UNIT_IN_HR = {'hr': 1.0, 'h': 1.0, 'hour': 1.0, 'min': 1.0 / 60.0, 's': 1.0 / 3600.0, 'day': 24.0, 'd': 24.0}


def GM_c3(t_units='hr', mass_msun=sgra_mass_msun):
    """GM/c^3 expressed in `t_units` (a units.Unit, a string, or None -> 1.0, emission.py:183-185)."""
    if t_units is None:
        return 1.0
    name = getattr(t_units, 'name', None) or str(t_units)
    if name not in _UNIT_IN_HR:
        raise AttributeError('time unit {} not supported'.format(name))
    return GM_c3_hr * (mass_msun / sgra_mass_msun) / _UNIT_IN_HR[name]
