# Not a test.  FETCH_SIZE of the throughput tracking kernel per launch, two libraries (GYPSUM_HIP_LIB) x three stream counts: one, two and three
# rounds of workgroups on the chip's 512 slots.  [FETCH_AB_STREAMS="42 85 128"] bash tools/fetch_ab_visit.sh <tag> <lib> [<lib> ...]   (libgypsum_<lib>.so under gypsum_amd/csrc)
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/${1:-fetch_ab}; mkdir -p $O
for streams in ${FETCH_AB_STREAMS:-42 85 128}; do
for lib in ${@:2}; do
    rm -rf /tmp/pm
    GYPSUM_HIP_LIB=$PWD/gypsum_amd/csrc/libgypsum_$lib.so timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pm -o bench -- python bench.py --streams $streams ${FETCH_AB_ARGS:-} --no-cpu-baseline --no-extras --no-telemetry --steps 2 --warmup 1 > /tmp/pm.log 2>&1
    python - $lib $streams <<'PY'
import csv,glob,sys,collections
d=collections.defaultdict(list); t=collections.defaultdict(list)
for f in glob.glob('/tmp/pm/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'track_block_kernel<8, false, 0>' in r['Kernel_Name']:
            d[r['Kernel_Name'][:40]].append(float(r['Counter_Value'])); t[r['Kernel_Name'][:40]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))*1e-6)
import os
streams=int(sys.argv[2]); launch_ms=float(os.environ.get("FETCH_AB_LAUNCH_MS", "500")); alg=streams*launch_ms*8184*8/1024
for k,v in d.items(): print(f"lib {sys.argv[1]:6s} {os.environ.get('FETCH_AB_ARGS', '')} streams {streams:3d} ({streams*12} workgroups), {launch_ms:.0f} ms of signal per launch: {len(v)} launches, FETCH_SIZE mean {sum(v)/len(v)/1e6:.3f} GB counted = {sum(v)/len(v)/alg:.3f} x the samples' bytes (min {min(v)/alg:.3f}, max {max(v)/alg:.3f}); {sum(t[k])/len(t[k]):.2f} ms per launch")
PY
done; done 2>&1 | tee $O/fetch_ab.txt
