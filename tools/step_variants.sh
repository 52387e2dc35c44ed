set -u
run() { tag=$1; shift; python bench.py --full-only --no-fp8 --no-cpu-baseline "$@" > gpurun_out/r02k_$tag.json 2> gpurun_out/r02k_$tag.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r02k_$tag.json")); print("$tag", d["value"], "img/s", d["ms_per_step"], "ms", d["peak_mem_gb"], "GB", d["phases_ms"])
except Exception as e:
    print("$tag FAILED", e); print(open("gpurun_out/r02k_$tag.err").read()[-800:])
PY
}
run batch1 --batch 1 --steps 20
run batch2 --batch 2 --steps 16
run batch16 --batch 16 --steps 6
run refsched --act-checkpoint on --match-twice --steps 8
run f32 --act-dtype f32 --steps 6
