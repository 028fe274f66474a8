import sys, time
sys.path.insert(0, '.')
import numpy as np
from gypsum_amd.engine import GypsumEngine
from gypsum_amd._lib import CHAN_INIT
eng = GypsumEngine(0); eng.set_stream_format(8_184_000, 8184)
a = np.zeros(1 << 28, dtype=np.uint8)  # 256 MB pageable
buf = eng.alloc(a.nbytes)
for _ in range(2):
    t = time.perf_counter(); buf.upload(a); dt = time.perf_counter() - t
print(f"pageable H2D 256 MB: {a.nbytes / dt / 1e9:.1f} GB/s")
# host-buffer form of the block tracker: 4 streams x 250 ms, 12 channels each
rng = np.random.default_rng(0)
B, T, n = 4, 250, 8184
iq = (rng.standard_normal(B * T * n) + 1j * rng.standard_normal(B * T * n)).astype(np.complex64)
inits = np.zeros(B * 12, dtype=CHAN_INIT)
for i in range(B * 12):
    inits[i] = (i // 12, i % 12 + 1, 100.0 * i, 0.0, 17 * i, 0)
bank = eng.create_bank(inits)
t0 = [round(ms * n / 8_184_000, 6) for ms in range(T)]
for _ in range(2):
    t = time.perf_counter(); bank.track_block(iq, B, T, t0); dt = time.perf_counter() - t
print(f"host-buffer gyp_track_block, {B} streams x {T} ms: {B * T * n / dt / 1e6:.0f} Msamples/s incl. H2D of IQ and D2H of records")
