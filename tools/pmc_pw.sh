#!/bin/bash
# SQ counters of the pointwise kernel (and the tiled kernel beside it) on one MobileOne shape: where a wave's cycles go.
# usage: tools/pmc_pw.sh OUTDIR CASE      (one counter group per pass, rocprofv3 --pmc with --kernel-trace only)
set -e
OUT=$1; CASE=$2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE GRBM_TA_BUSY"; do
  i=$((i + 1))
  tag=$(printf "p%02d" $i)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/$tag" -o p -- python3 "$GRAFT_REPO_ROOT/tools/pw_lab.py" --cases "$CASE" --only pw,0 --iters 2 --reps 2 > "$GRAFT_REPO_ROOT/$OUT/$tag.log" 2>&1 || echo "pass $tag ($grp) failed"
  echo "pass $tag done"
done
cd "$GRAFT_REPO_ROOT"
python3 tools/pmc_summary.py $OUT/p* > $OUT/summary.json 2>/dev/null || true
