"""GPS L1 C/A Gold codes (host mirror of the reference's `gypsum/gps_ca_prn_codes.py`).

Same public names as the reference (`GpsSatelliteId`, `GpsReplicaPrnSignal`,
`generate_replica_prn_signals`, gps_ca_prn_codes.py:33-52,134-250) so receiver-side
code can import either.  The device library generates the identical table
internally (`gyp_prn_chips`); `tests/test_abi_and_host.py` checks the two against each
other, against the reference's own table (`tests/golden/prn_chips.npz`) and against the
frozen sha256 of SURVEY section 8(c5).

Generator: two 10-stage LFSRs, both seeded all-ones.  G1 = x^10 + x^3 + 1,
G2 = x^10 + x^9 + x^8 + x^6 + x^3 + x^2 + 1; chip = G1[10] ^ G2[tap_a] ^ G2[tap_b]
(IS-GPS-200 table 3-Ia).  Registers are held as 10-bit integers here (bit i-1 =
stage i) rather than Python lists.
"""
from __future__ import annotations

from dataclasses import dataclass
from functools import lru_cache
from typing import Any, Dict

import numpy as np

PRN_CHIP_COUNT = 1023

# (tap_a, tap_b) of the G2 register for SV 1..32
G2_OUTPUT_TAPS = (
    (2, 6), (3, 7), (4, 8), (5, 9), (1, 9), (2, 10), (1, 8), (2, 9), (3, 10), (2, 3), (3, 4), (5, 6), (6, 7),
    (7, 8), (8, 9), (9, 10), (1, 4), (2, 5), (3, 6), (4, 7), (5, 8), (6, 9), (1, 3), (4, 6), (5, 7), (6, 8),
    (7, 9), (8, 10), (1, 6), (2, 7), (3, 8), (4, 9),
)
# IS-GPS-200 "first 10 chips, octal" column, SV 1..32
FIRST_TEN_CHIPS_OCTAL = (
    0o1440, 0o1620, 0o1710, 0o1744, 0o1133, 0o1455, 0o1131, 0o1454, 0o1626, 0o1504, 0o1642, 0o1750, 0o1764,
    0o1772, 0o1775, 0o1776, 0o1156, 0o1467, 0o1633, 0o1715, 0o1746, 0o1763, 0o1063, 0o1706, 0o1743, 0o1761,
    0o1770, 0o1774, 0o1127, 0o1453, 0o1625, 0o1712,
)


@dataclass
class GpsSatelliteId:
    id: int

    def __init__(self, id: int) -> None:
        self.id = id

    def __hash__(self) -> int:
        return hash(self.id)

    def __eq__(self, other: Any) -> bool:
        return isinstance(other, type(self)) and self.id == other.id


@dataclass
class GpsReplicaPrnSignal:
    inner: np.ndarray


def _stage(reg: int, i: int) -> int:
    return (reg >> (i - 1)) & 1


@lru_cache(maxsize=1)
def generate_ca_code_table() -> np.ndarray:
    """uint8[32, 1023] chips in {0,1}.  Raises ValueError if any code misses its IS-GPS-200 marker."""
    g1 = g2 = 0x3FF
    table = np.zeros((32, PRN_CHIP_COUNT), dtype=np.uint8)
    for c in range(PRN_CHIP_COUNT):
        o1 = _stage(g1, 10)
        for sv, (ta, tb) in enumerate(G2_OUTPUT_TAPS):
            table[sv, c] = o1 ^ _stage(g2, ta) ^ _stage(g2, tb)
        fb1 = _stage(g1, 3) ^ _stage(g1, 10)
        fb2 = (_stage(g2, 2) ^ _stage(g2, 3) ^ _stage(g2, 6) ^ _stage(g2, 8) ^ _stage(g2, 9) ^ _stage(g2, 10))
        g1 = ((g1 << 1) & 0x3FF) | fb1
        g2 = ((g2 << 1) & 0x3FF) | fb2
    for sv in range(32):
        head = 0
        for c in range(10):
            head = (head << 1) | int(table[sv, c])
        if head != FIRST_TEN_CHIPS_OCTAL[sv]:
            raise ValueError(f"SV {sv + 1}: generated PRN starts {head:o}, IS-GPS-200 says {FIRST_TEN_CHIPS_OCTAL[sv]:o}")
    table.setflags(write=False)
    return table


def generate_replica_prn_signals() -> Dict[GpsSatelliteId, GpsReplicaPrnSignal]:
    """Same contract as gps_ca_prn_codes.py:134: {GpsSatelliteId(i): GpsReplicaPrnSignal(int64[1023] in {0,1})}."""
    table = generate_ca_code_table()
    return {GpsSatelliteId(sv + 1): GpsReplicaPrnSignal(table[sv].astype(np.int64)) for sv in range(32)}
