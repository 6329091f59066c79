#!/bin/bash
# SQ counter passes on the chain kernel through tools/chain_lab.py (lab library).  usage: tools/pmc_chain.sh OUTDIR CASES
# One counter group per pass (rocprofv3 --pmc with --kernel-trace only), summarised by tools/pmc_summary.py.
set -e
OUT=$1; CASES=$2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/$tag" -o p -- python3 "$GRAFT_REPO_ROOT/tools/chain_lab.py" --cases "$CASES" --rows 64 --iters 3 > "$GRAFT_REPO_ROOT/$OUT/$tag.log" 2>&1 || echo "pass $tag failed"
done
cd "$GRAFT_REPO_ROOT"
python3 tools/pmc_summary.py $OUT/* > $OUT/summary.json 2>/dev/null || true
