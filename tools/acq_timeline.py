"""Not a test.  `python tools/acq_timeline.py run` = three 13-stream 32-satellite scans at 8.184 Msps (run it under
`rocprofv3 --kernel-trace --output-format csv -d DIR -o acq`); `python tools/acq_timeline.py show DIR` prints the LAST scan's
kernel timeline from the trace: per launch the start offset, duration and the idle gap before it, plus totals per kernel."""
import collections
import csv
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def run():
    import numpy as np

    from bench import ALL_IDS, make_scene
    from gypsum_amd._lib import ACQ_RESULT
    from gypsum_amd.engine import GypsumEngine

    fs, n, A, T = 8_184_000, 8184, 13, 10
    eng = GypsumEngine(0)
    eng.set_stream_format(fs, n)
    rng = np.random.default_rng(3)
    scene = make_scene(rng, A, 12, fs, 0.005)
    iq = eng.alloc(A * T * n * 8)
    eng.synth_iq(iq, A, T * n, T, scene, 0.03, 99)
    out = eng.alloc(A * 32 * ACQ_RESULT.itemsize)
    for i in range(3):
        eng.timer_start()
        eng.acquire_dev(iq.ptr.value, A, T * n, 10, ALL_IDS, out.ptr.value)
        print(f"scan {i}: {eng.timer_stop():.3f} ms")
    eng.close()


def show(d):
    f = next(Path(d).rglob("*kernel_trace.csv"))
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    # the last scan starts at the last acq_plan_kernel that follows an acq_finish_kernel (or the first one)
    starts = [i for i, r in enumerate(rows) if "acq_plan_kernel" in r["Kernel_Name"] and (i == 0 or "acq_finish" in rows[i - 1]["Kernel_Name"]
                                                                                             or "synth" in rows[i - 1]["Kernel_Name"])]
    if not starts:   # r04 on: a scan's search states are initialised on the device -- it starts with one acq_init_kernel per part (two parts)
        inits = [i for i, r in enumerate(rows) if "acq_init_kernel" in r["Kernel_Name"]]
        starts = [inits[-2] if len(inits) >= 2 else inits[-1]]
    rows = rows[starts[-1]:]
    t0 = int(rows[0]["Start_Timestamp"])
    prev_end = t0
    tot = collections.defaultdict(lambda: [0, 0.0])
    gaps = 0.0
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("gyp::", "").replace("void ", "").split("(")[0]
        print(f"{(s - t0) / 1e3:10.1f} us  +{(s - prev_end) / 1e3:7.1f} gap  {(e - s) / 1e3:9.1f} us  {name}  grid {r.get('Grid_Size_X', '?')}")
        tot[name][0] += 1
        tot[name][1] += (e - s) / 1e3
        gaps += max(0, s - prev_end) / 1e3
        prev_end = max(prev_end, e)
    print(f"--- scan: {(prev_end - t0) / 1e3:.1f} us, idle gaps {gaps:.1f} us over {len(rows)} launches")
    for k, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print(f"{t:10.1f} us  {c:4d} x  {k}")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else show(sys.argv[2])
