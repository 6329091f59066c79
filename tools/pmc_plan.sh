#!/bin/bash
# Per-kernel instruction counters (SQ_INSTS_VALU / SALU / LDS / MFMA, waves, issue-wait share) of one forward of a frozen plan:
# vector instructions per launch x 4 clocks / 1024 SIMDs is the time the launch needs for its vector instructions alone (round 4:
# the quantising epilogues are bound by exactly that).  usage: tools/pmc_plan.sh MODEL BATCH OUTDIR   (prints the last forward's launches)
M=$1; B=$2; OUT=$3
mkdir -p $GRAFT_REPO_ROOT/$OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/$OUT/p1 -o p -- python3 $GRAFT_REPO_ROOT/tools/plan_profile.py $M $B > $GRAFT_REPO_ROOT/$OUT/p1.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - "$OUT" <<'PY'
import csv,sys,glob,re,collections
out=sys.argv[1]
path=glob.glob(out+"/p1/**/*counter_collection.csv",recursive=True)[0]
rows=list(csv.DictReader(open(path)))
by=collections.OrderedDict()
for r in rows:
    k=(int(r["Dispatch_Id"]), re.sub(r"\(.*","",r["Kernel_Name"]).replace("void ","")[:70], r.get("Grid_Size",""), r.get("Workgroup_Size",""))
    by.setdefault(k,{})[r["Counter_Name"]]=float(r["Counter_Value"])
# last forward only: take the last N dispatches where N = dispatches per forward: print the last 60
keys=list(by)[-64:]
for k in keys:
    v=by[k]
    print(f"{k[0]:6d} {k[1]:70s} grid {k[2]:>9s} wg {k[3]:>5s} VALU {v.get('SQ_INSTS_VALU',0)/1e6:8.2f}M SALU {v.get('SQ_INSTS_SALU',0)/1e6:7.2f}M LDS {v.get('SQ_INSTS_LDS',0)/1e6:7.2f}M MFMA {v.get('SQ_INSTS_MFMA',0)/1e6:6.2f}M waves {v.get('SQ_WAVES',0):8.0f} wait_inst {v.get('SQ_WAIT_INST_ANY',0)/max(v.get('SQ_WAVE_CYCLES',1),1):.2f}")
PY
find "$GRAFT_REPO_ROOT/$OUT/p1" -name "*.csv" -delete 2>/dev/null || true
