#!/bin/bash
# Evidence run for profiles/: tests, bench JSON, rocprofv3 kernel stats and the two PMC passes (separate runs, as
# MI355X_MICROARCH.md prescribes).  usage (on the GPU box, from the repo root): bash tools/profile_round.sh r01d
set -u
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python "$R/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err"
B="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-trunk --no-roofline"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o bench -- $B > "$OUT/bench_stats.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o bench -- $B > "$OUT/bench_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -o bench -- $B > "$OUT/bench_write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d "$OUT/pmc_mfma" -o bench -- $B > "$OUT/bench_mfma.log" 2>&1
cd "$R"
S=$(find "$OUT/stats" -name '*.db' | head -1); F=$(find "$OUT/pmc_fetch" -name '*.db' | head -1); W=$(find "$OUT/pmc_write" -name '*.db' | head -1)
python tools/rocpd_summary.py stats "$S" > "$OUT/${TAG}_kernel_stats.txt"
python tools/rocpd_summary.py pmc "$F" > "$OUT/${TAG}_pmc_fetch.txt"
python tools/rocpd_summary.py pmc "$W" > "$OUT/${TAG}_pmc_write.txt"
python tools/rocpd_summary.py traffic "$F" "$W" > "$OUT/${TAG}_traffic.json"
Q=$(find "$OUT/pmc_mfma" -name '*.db' | head -1); [ -n "$Q" ] && python tools/rocpd_summary.py pmc "$Q" > "$OUT/${TAG}_pmc_mfma.txt"
cp "$OUT/bench.json" "$OUT/${TAG}_bench.json"
rm -rf "$OUT/stats" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_mfma"      # the sqlite files are large; the summaries are what is kept
tail -c 600 "$OUT/bench.json"; echo; head -12 "$OUT/${TAG}_kernel_stats.txt"
