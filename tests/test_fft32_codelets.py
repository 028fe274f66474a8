"""The generated 32-point DFT codelets (tools/gen_fft32.py -> gypsum_amd/csrc/fft32_gen.hpp): evaluated in float64 against
numpy.fft, and the committed header must be what the generator writes."""
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]


def test_generated_codelets_match_numpy_fft():
    r = subprocess.run([sys.executable, str(REPO / "tools" / "gen_fft32.py"), "--check"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.count("388 float instructions") == 3, r.stdout


def test_committed_header_is_the_generators_output(tmp_path):
    sys.path.insert(0, str(REPO / "tools"))
    import gen_fft32
    want = (REPO / "gypsum_amd" / "csrc" / "fft32_gen.hpp").read_text()
    keep = gen_fft32.OUT
    gen_fft32.OUT = tmp_path / "fft32_gen.hpp"
    try:
        argv, sys.argv = sys.argv, ["gen_fft32.py"]
        gen_fft32.main()
    finally:
        sys.argv = argv
        gen_fft32.OUT = keep
    assert (tmp_path / "fft32_gen.hpp").read_text() == want
