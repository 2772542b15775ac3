# usage: gpu_pmc_variant.sh "<extra -D flags>" "<bench args>" <tag>: rebuild kernel4 with flags, count VALU instructions per launch
cd $GRAFT_REPO_ROOT
REPO=$GRAFT_REPO_ROOT
CSRC=mod-interleaveavx_multithreads-faid_amd/csrc
trap 'make -s -C $REPO/$CSRC EXTRA= > $REPO/gpurun_out/pmcv_restore.build.log 2>&1' EXIT   # always end on the default build
make -s -C $CSRC EXTRA="$1" > gpurun_out/pmcv_$3.build.log 2>&1 || { echo build failed; exit 1; }
OUT=$REPO/gpurun_out/pmcv_$3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT -o p -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-points --no-cpu $2 > $OUT/bench.json 2> $OUT/err.log
python3 - <<PY
import csv, collections
rows=[r for r in csv.DictReader(open("$OUT/p_counter_collection.csv")) if "lnsfaid_decode" in r["Kernel_Name"]]
by=collections.OrderedDict()
for r in rows:
    by.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]]=float(r["Counter_Value"])
tot=collections.Counter()
for d,c in by.items():
    print("$3 dispatch", d, {k: round(v/1e9,3) for k,v in c.items()})
    for k,v in c.items(): tot[k]+=v
print("$3 TOTAL", {k: round(v/1e9,3) for k,v in tot.items()})
PY
