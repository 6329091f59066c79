#!/bin/bash
# Collects the round's measurement artefacts on the GPU box into gpurun_out/prof_rNN/ (copy what is to be judged into profiles/).
# usage: tools/collect_profiles.sh r02
set -e
R=${1:-rNN}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$R
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py"
# PART=1: bench lines, kernel stats, HBM traffic, plan profiles; PART=2: lab tools and counter passes; unset: everything (each part fits one
# gpurun call of 20 minutes)
PART=${PART:-all}
if [ "$PART" = all ] || [ "$PART" = 1 ]; then
python3 "$B" --steps 20 --warmup 5 > "$OUT/bench_line.json" 2> "$OUT/bench.err"
python3 "$B" --steps 20 --warmup 5 --streams 1 --no-cpu-baseline > "$OUT/bench_line_streams1.json" 2>> "$OUT/bench.err"
# per-kernel durations of the same command (one stream: rocprofv3 and the HIP events then describe the same thing)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o b -- python3 "$B" --steps 20 --warmup 5 --streams 1 --no-cpu-baseline --no-other-configs --no-fake-quant-leg > "$OUT/stats.log" 2>&1 || echo "stats pass failed"
# HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md, HBM section).  --no-other-configs / --no-fake-quant-leg:
# `pmc_summary.py --tail KERNEL N` takes the LAST N dispatches of a kernel in the process, so nothing but the headline network may run in
# these passes (round 4's passes also ran RepVGG-A1 and MobileOne-S1 behind it: two families' tails were the side networks' launches)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o p -- python3 "$B" --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-other-configs --no-fake-quant-leg > "$OUT/fetch.log" 2>&1 || echo "fetch pass failed"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o p -- python3 "$B" --steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-other-configs --no-fake-quant-leg > "$OUT/write.log" 2>&1 || echo "write pass failed"
cd "$GRAFT_REPO_ROOT"
# (several tails per kernel where the per-step launch count depends on the dispatch: bench.py takes the one whose count is its step's)
python3 tools/pmc_summary.py --tail conv_chain_i8_kernel 11 --tail conv_i8_mfma_kernel 4 --tail conv_i8_mfma_kernel 5 --tail conv_i8_mfma_kernel 6 --tail conv3x3_halo_i8_kernel 16 --tail conv_pwr_i8_kernel 4 --tail conv_pw_i8_kernel 3 --tail conv_stem_pool7_i8_kernel 1 --tail quantize_pad_nhwc4 1 "$OUT/fetch" "$OUT/write" > "$OUT/pmc_bench_fused_plan.json" || true
# the side networks' own passes (their rooflines' traffic): RepVGG-A1's 21 halo-kernel layers, MobileOne-S1's depthwise / pointwise families
cd /tmp
for mk in "repvgg_a1 conv3x3_halo_i8_kernel:19,conv3x3_halo_i8_kernel:20,conv3x3_halo_i8_kernel:21" "mobileone_s1 conv_dwm_i8_kernel:17,conv_pw_i8_kernel:20,conv_pw_i8_kernel:21,conv_dw3_i8_kernel:4"; do
  set -- $mk; M=$1; TAILS=""
  for kv in $(echo $2 | tr ',' ' '); do TAILS="$TAILS --tail ${kv%%:*} ${kv##*:}"; done
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch_$M" -o p -- python3 "$B" --model $M --steps 3 --warmup 1 --streams 1 --no-cpu-baseline > "$OUT/fetch_$M.log" 2>&1 || echo "fetch pass $M failed"
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write_$M" -o p -- python3 "$B" --model $M --steps 3 --warmup 1 --streams 1 --no-cpu-baseline > "$OUT/write_$M.log" 2>&1 || echo "write pass $M failed"
  (cd "$GRAFT_REPO_ROOT" && python3 tools/pmc_summary.py $TAILS "$OUT/fetch_$M" "$OUT/write_$M" > "$OUT/pmc_bench_$M.json") || true
done
cd "$GRAFT_REPO_ROOT"
python3 tools/plan_profile.py resnet50 512 > "$OUT/plan_profile_resnet50_b512.txt" 2>&1
python3 tools/plan_profile.py repvgg_a1 512 > "$OUT/plan_profile_repvgg_a1_b512.txt" 2>&1
python3 tools/plan_profile.py mobileone_s1 1024 > "$OUT/plan_profile_mobileone_s1_b1024.txt" 2>&1
python3 "$B" --model repvgg_a1 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_line_repvgg_a1.json" 2>> "$OUT/bench.err"
python3 "$B" --model mobileone_s1 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_line_mobileone_s1.json" 2>> "$OUT/bench.err"
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/bench_kernel_stats.csv" \;
fi
if [ "$PART" = all ] || [ "$PART" = 2 ]; then
cd "$GRAFT_REPO_ROOT"
# the halo-tile 3x3 kernel (round 3): ablations + per-workgroup clock stamps, and SQ / TCC counters of the kernel alone
python3 tools/halo_lab.py --cases c1,c2,c3,c4 --generic --stamps > "$OUT/halo_lab_resnet50_3x3.txt" 2>&1
bash tools/pmc_layer.sh "gpurun_out/prof_$R/pmc_c3_halo" c3 128:5 > /dev/null 2>&1 && cp "$OUT/pmc_c3_halo/summary.json" "$OUT/pmc_conv3x3_256_14.json" || echo "pmc c3 failed"
bash tools/pmc_layer.sh "gpurun_out/prof_$R/pmc_c4_halo" c4 128:5 > /dev/null 2>&1 && cp "$OUT/pmc_c4_halo/summary.json" "$OUT/pmc_conv3x3_512_7.json" || echo "pmc c4 failed"
python3 tools/first_batch_probe.py > "$OUT/first_batch_probe.txt" 2>&1
python3 tools/conv_lab.py --knobs 128:1,64:1,128:0,64:0,128:5,64:5 > "$OUT/conv_lab_resnet50_layers.txt" 2>&1
python3 tools/chain_lab.py --rows 64 > "$OUT/chain_lab_resnet50_pairs.txt" 2>&1
# where the chain kernel's vector-memory instructions wait: SQ / TA / TCP / TCC / TD counter passes (round 4)
bash tools/pmc_chain_mem.sh "gpurun_out/prof_$R/pmc_chain_mem" s3,s2t,s1,s2 > /dev/null 2>&1 && cp "$OUT/pmc_chain_mem/summary.json" "$OUT/pmc_chain_kernel_sq_ta_tcp_tcc.json" || echo "pmc chain failed"
python3 tools/chain_trace.py 512 64 56 256 64 > "$OUT/chain_trace_64_256_64_at_56.txt" 2>&1
python3 tools/chain_trace.py 512 256 14 1024 256 > "$OUT/chain_trace_256_1024_256_at_14.txt" 2>&1
python3 tools/conv_trace.py 512 256 14 256 3 > "$OUT/conv_trace_3x3_256_14.txt" 2>&1
# round 4: the pointwise kernel beside the tiled one (ablations, one wave's stamps), its SQ counters, plan-wide instruction counters,
# the vector-instruction issue-rate probe
python3 tools/pw_lab.py --trace > "$OUT/pw_lab_mobileone_pointwise.txt" 2>&1
bash tools/pmc_pw.sh "gpurun_out/prof_$R/pmc_pw" 192x28 > /dev/null 2>&1 && cp "$OUT/pmc_pw/summary.json" "$OUT/pmc_pointwise_192_28_sq.json" || echo "pmc pw failed"
for mb in "resnet50 512" "repvgg_a1 512" "mobileone_s1 1024"; do set -- $mb; bash tools/pmc_plan.sh $1 $2 "gpurun_out/prof_$R/pmc_plan_$1" > "$OUT/pmc_plan_instruction_counts_$1.txt" 2>&1 || echo "pmc plan $1 failed"; done
/opt/rocm/bin/hipcc -O2 -Wno-unused-value --offload-arch=gfx950 tools/probes/valu_probe.hip -o /tmp/valu_probe 2>/dev/null && timeout -k 5 60 /tmp/valu_probe > "$OUT/valu_issue_probe.txt" 2>&1 || echo "valu probe failed"
python3 tools/kernel_bench.py > "$OUT/kernel_bench_config2.txt" 2>&1
python3 tools/run_configs.py > "$OUT/configs_1_3_4_5.json" 2> "$OUT/configs.err"
fi
# the raw per-dispatch tables are large (gpurun copies at most 64 MiB back): the summaries above are what is kept
find "$OUT" -mindepth 2 \( -name "*counter_collection.csv" -o -name "*kernel_trace.csv" -o -name "*agent_info.csv" \) -delete
ls -la "$OUT"
