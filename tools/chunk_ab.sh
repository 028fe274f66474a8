export GYP_TEST_HOOKS=1   # GypsumEngine forwards GYP_* switches (gyp_debug_set) only under this opt-in
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for C in 0 500 250 125; do
  echo "== GYP_TRACK_CHUNK_MS=$C"
  GYP_TRACK_CHUNK_MS=$C timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys
l=json.loads(sys.stdin.read()); print({k:l[k] for k in ('value','ms_per_step','acquire_ms_per_step','track_ms_per_step')}, l['track_kernels_ms_per_step'])"
  GYP_TRACK_CHUNK_MS=$C timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r03_chunk_$C -o b -- python bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/r03_chunk_$C/**/*counter_collection.csv", recursive=True)[0]
v=collections.defaultdict(float); n=collections.Counter()
seen=set()
for r in csv.DictReader(open(f)):
    if "track_block_kernel" in r["Kernel_Name"]:
        v["t"]+=float(r["Counter_Value"]); seen.add(r["Dispatch_Id"])
print("launches",len(seen),"total FETCH_SIZE KB over",len(seen),"launches:",v["t"])
PY
  rm -rf gpurun_out/r03_chunk_$C
done
