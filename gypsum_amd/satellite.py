"""`GpsSatellite` -- host mirror of the reference's `gypsum/satellite.py:8-31`."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .gps_ca_prn_codes import GpsReplicaPrnSignal, GpsSatelliteId

ALL_SATELLITE_IDS = [GpsSatelliteId(i + 1) for i in range(32)]


@dataclass
class GpsSatellite:
    satellite_id: GpsSatelliteId
    prn_code: GpsReplicaPrnSignal
    scale_factor: int

    def __hash__(self) -> int:
        return hash(self.satellite_id)

    @property
    def prn_as_complex(self) -> np.ndarray:
        """+-1 replica, `scale_factor` samples per chip, complex128 (satellite.py:20-31)."""
        cached = self.__dict__.get("_prn_as_complex")
        if cached is None:
            chips = np.asarray(self.prn_code.inner).astype(np.int64)
            cached = (2 * np.repeat(chips, self.scale_factor) - 1).astype(complex)
            self.__dict__["_prn_as_complex"] = cached
        return cached
