#!/bin/bash
# The memory path of the pointwise kernel (conv_pw_i8_kernel) on one MobileOne shape: requests, round trips, the L1's pending stalls
# (round 5, LABNOTES 16: is it on the same per-CU read-slot bound as the chain kernel?).  usage: tools/pmc_pw_mem.sh OUTDIR CASE
set -e
OUT=$1; CASE=$2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ_LATENCY TCP_TCC_WRITE_REQ" \
           "TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES" \
           "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_NC_READ_REQ_sum" \
           "TCC_HIT TCC_MISS TCC_REQ TCC_STREAMING_REQ"; do
  i=$((i + 1))
  tag=$(printf "p%02d" $i)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/$tag" -o p -- python3 "$GRAFT_REPO_ROOT/tools/pw_lab.py" --cases "$CASE" --only pw,0 --iters 2 --reps 2 > "$GRAFT_REPO_ROOT/$OUT/$tag.log" 2>&1 || echo "pass $tag ($grp) failed"
done
cd "$GRAFT_REPO_ROOT"
python3 tools/pmc_summary.py $OUT/p* > $OUT/summary.json 2>/dev/null || true
python3 - <<PY
import json
d=json.load(open("$OUT/summary.json"))
for k,v in d.items():
    if "conv_pw" in k:
        print(k[:70]); [print("   %-40s %.1f" % (c, x["mean_KiB"])) for c,x in sorted(v.items())]
PY
