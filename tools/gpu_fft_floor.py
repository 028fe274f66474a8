"""Not a test: measures the transform-pair floor (debug hook gyp_debug_fft_bench)."""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from gypsum_amd.engine import GypsumEngine  # noqa: E402

eng = GypsumEngine(0)
eng.set_stream_format(8_184_000, 8184)
iters = 400
for waves, wgs_per_cu in ((1, 1), (4, 1), (8, 1), (8, 2), (4, 3), (4, 4), (2, 8), (1, 16)):
    ms = C.c_float()
    wgs = 256 * wgs_per_cu
    eng._check(eng.lib.gyp_debug_fft_bench(eng.ctx, waves, wgs, iters, C.byref(ms)))
    per_cu_waves = waves * wgs_per_cu
    us_per_pair_wave = ms.value * 1e3 / iters
    print(f"waves/WG {waves} WGs/CU {wgs_per_cu} ({per_cu_waves} waves/CU): {ms.value:.3f} ms -> {us_per_pair_wave:.3f} us per transform pair per wave; "
          f"CU throughput {per_cu_waves / us_per_pair_wave:.2f} pairs/us; chip {256 * per_cu_waves / us_per_pair_wave / 1e3:.2f} Gpairs/s... "
          f"cycles@2.4GHz per pair per wave {us_per_pair_wave * 2400:.0f}")
