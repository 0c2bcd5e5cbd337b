"""Migration library: same five names as the reference's
``src/impdar/lib/migrationlib/__init__.py:13-19``, all backed by the HIP
engine.  Unlike the reference there is no Python fallback: if the HIP
extension cannot be loaded the calls raise (``impdar_amd._hip.HipUnavailableError``).
"""
from .mig_hip import (migrationKirchhoff, migrationStolt, migrationPhaseShift,
                      migrationTimeWavenumber, getVelocityProfile)
from .mig_su import migrationSeisUnix

__all__ = ['migrationKirchhoff', 'migrationStolt', 'migrationPhaseShift',
           'migrationTimeWavenumber', 'migrationSeisUnix', 'getVelocityProfile']
