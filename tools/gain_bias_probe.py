"""Not a test: the relative magnitude error of the device's float32 correlation peak against float64, over Doppler, for a noise-free
one-satellite millisecond -- is there a SYSTEMATIC gain error in the float32 correlator (r06: a gain bias on the prompt peak perturbs
the Costas loop's gain, which an unlocked loop amplifies)?    python tools/gain_bias_probe.py <fs>"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from gypsum_amd._lib import CELL_DESC, GYP_COHERENT  # noqa: E402
from gypsum_amd.engine import GypsumEngine  # noqa: E402
from oracle import gypsum_oracle as orc  # noqa: E402

fs = int(sys.argv[1]) if len(sys.argv) > 1 else 8_184_000
n = fs // 1000
eng = GypsumEngine(0)
eng.set_stream_format(fs, n)
chips = orc.generate_ca_codes()
rng = np.random.default_rng(5)
t = np.arange(n) / fs
for label, noise in (("noise-free", 0.0), ("sigma = 6 a", 6.0)):
    out = []
    for trial in range(48):
        sv = int(rng.integers(1, 33))
        d = 0.0 if trial < 8 else float(rng.uniform(-4500, 4500))
        cp = int(rng.integers(0, n))
        prn = orc.prn_as_complex(chips[sv - 1], n).real
        a = 20.0 / n
        x = a * np.roll(prn, cp) * np.exp(1j * (2 * np.pi * d * t + rng.uniform(0, 6.28)))
        if noise:
            x = x + noise * a * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
        iq = x.astype(np.complex64)
        cell = np.zeros(1, dtype=CELL_DESC)
        cell[0] = (0, sv, d, -1, 0)
        _, prof = eng.correlate_cells(iq, 1, 1, cell, GYP_COHERENT, want_profiles=True)
        dev = prof[0][cp]
        ref = np.sum(iq.astype(np.complex128) * np.exp(-1j * (2 * np.pi * d * t)) * np.roll(prn, cp))
        out.append((d, abs(dev) / abs(ref) - 1.0, float(np.angle(dev * np.conj(ref)))))
    out = np.array(out)
    z, nz = out[:8], out[8:]
    print(f"{fs / 1e6:.3f} Msps, {label}: relative magnitude error at Doppler 0: mean {z[:, 1].mean():+.2e} (std {z[:, 1].std():.1e}); at random Doppler: mean {nz[:, 1].mean():+.2e} "
          f"(std {nz[:, 1].std():.1e}, min {nz[:, 1].min():+.2e}, max {nz[:, 1].max():+.2e}); rotation mean {nz[:, 2].mean():+.2e} rad (std {nz[:, 2].std():.1e})")
