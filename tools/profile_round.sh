#!/bin/bash
# Evidence run for profiles/: bench JSON, rocprofv3 kernel stats of the adapter path and of the whole step (steady state),
# and the PMC passes over the adapter path (separate runs, as MI355X_MICROARCH.md prescribes).
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh r02c
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python "$R/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
A="python $R/bench.py --adapter-only --steps 3 --warmup 1 --no-cpu-baseline --no-roofline"
F="python $R/bench.py --full-only --steps 4 --warmup 2 --no-fp8 --no-cpu-baseline --no-fused-ab --no-literal --no-fp32-layout --no-parity --no-data-step"
rocprofv3 --kernel-trace --stats -d /tmp/p_stats -o bench -- $A > "$OUT/stats.log" 2>&1
rocprofv3 --kernel-trace --stats -d /tmp/p_full -o bench -- $F > "$OUT/full.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_fetch -o bench -- $A > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_write -o bench -- $A > "$OUT/write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d /tmp/p_mfma -o bench -- $A > "$OUT/mfma.log" 2>&1
B="python $R/tools/block_traffic_run.py"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_bfetch -o bench -- $B > "$OUT/bfetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_bwrite -o bench -- $B > "$OUT/bwrite.log" 2>&1
cd "$R"
db() { find "$1" -name '*.db' | head -1; }
python tools/rocpd_summary.py stats "$(db /tmp/p_stats)" > "$OUT/${TAG}_kernel_stats.txt"
python tools/rocpd_summary.py stats_all "$(db /tmp/p_full)" 60 naive_conv > "$OUT/${TAG}_fullstep_kernel_stats.txt"
python tools/rocpd_summary.py gaps "$(db /tmp/p_full)" > "$OUT/${TAG}_fullstep_idle.txt"
python tools/rocpd_summary.py pmc "$(db /tmp/p_fetch)" > "$OUT/${TAG}_pmc_fetch.txt"
python tools/rocpd_summary.py pmc "$(db /tmp/p_write)" > "$OUT/${TAG}_pmc_write.txt"
python tools/rocpd_summary.py traffic "$(db /tmp/p_fetch)" "$(db /tmp/p_write)" > "$OUT/${TAG}_traffic.json"
python tools/rocpd_summary.py block_traffic "$(db /tmp/p_bfetch)" "$(db /tmp/p_bwrite)" 24 > "$OUT/${TAG}_block_traffic.json"
Q=$(db /tmp/p_mfma); [ -n "$Q" ] && python tools/rocpd_summary.py pmc "$Q" > "$OUT/${TAG}_pmc_mfma.txt"
cp "$OUT/bench.json" "$OUT/${TAG}_bench.json"
rm -f "$OUT"/*.log
tail -c 400 "$OUT/bench.json"; echo; head -14 "$OUT/${TAG}_kernel_stats.txt"
