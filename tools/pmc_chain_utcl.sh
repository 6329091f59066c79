#!/bin/bash
# usage: utcl.sh OUTDIR [--cm]
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for grp in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" "TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum" "TCP_UTCL1_LFIFO_FULL_sum TCP_UTCL1_THRASHING_STALL_sum"; do
  d=$OUT/$(echo $grp | tr ' ' '_' | cut -c1-60)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -o p -- python3 $GRAFT_REPO_ROOT/tools/chain_ab.py --cases d1,s1,s1t,s3 --iters 3 "$@" > $d.log 2>&1 || echo "pass failed: $grp"
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r.get("Kernel_Name","")
        if "conv_chain" not in k: continue
        name=k.split("conv_chain_i8_kernel")[1][:22]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name in sorted(acc):
    print(name, {c: round(sum(v[-3:])/len(v[-3:])/1e6,3) for c,v in sorted(acc[name].items())}, "(M per launch)")
PY
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete
