"""Not a test: which (a*N, sigma^2 N) let a 12-SATELLITE stream lock?  The float64 oracle tracker (CPU) on scenes of twelve satellites:
fraction of milliseconds with is_locked() and the number of lock <-> unlock transitions per channel.  bench.py's lock-regime legs
(lock_regime_amplitudes: a*N = 16, sigma^2 N = 0.3) come from this table.
    python tools/lock_scene_probe.py <fs> <n_ms> [aN,aN,...] [v,v,...]"""
import os, sys
os.environ["OPENBLAS_NUM_THREADS"]="1"
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np, multiprocessing as mp
def run(args):
    fs, aN, v, nsat, seed, n_ms, ch = args
    from gypsum_amd import synth
    from oracle import gypsum_oracle as orc
    import survey_worker
    n = fs//1000
    scene = synth.random_scene(fs, n_ms, nsat, seed, max_code_phase=(2046 if n>2046 else None), amplitude=aN/n, noise_sigma=(v/n)**0.5)
    iq = synth.render(scene)
    rng = np.random.default_rng(seed ^ 0x5EED)
    inits = survey_worker.scene_inits(scene, rng)
    chips = orc.generate_ca_codes()
    sv,dop,phi,cp = inits[ch]
    trk = orc.Tracker(orc.TrackingState(dop,phi,cp), orc.prn_as_complex(chips[sv-1], n), fs, n)
    locked=[]
    for ms in range(9,n_ms):
        st,en = orc.chunk_times(ms*n,n,fs)
        r = trk.process_samples(iq[ms*n:(ms+1)*n], st, en)
        locked.append(r.locked)
    l=np.array(locked,dtype=int)
    return (aN,v,ch,l.mean(), int(np.abs(np.diff(l)).sum()))
if __name__=="__main__":
    fs=int(sys.argv[1]); n_ms=int(sys.argv[2])
    grid_a = [float(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else (10, 12, 14, 16)
    grid_v = [float(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else (0.2, 0.5)
    jobs=[(fs,aN,v,12,77,n_ms,ch) for aN in grid_a for v in grid_v for ch in range(4)]
    with mp.Pool(8) as p:
        for r in p.imap(run,jobs): print(r,flush=True)
