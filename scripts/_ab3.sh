BASE="--steps 3 --warmup 1 --no-cpu-baseline --no-accuracy --no-dropin --no-single --no-regime-b --no-mng"
run() { name=$1; shift; timeout 400 python bench.py $BASE "$@" > gpurun_out/ab3_$name.json 2>/dev/null; python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.loads(open("gpurun_out/ab3_%s.json"%n).read().strip().splitlines()[-1])
    print("%-28s pc %.2f ms"%(n, d["post_correct"]["seconds"]*1e3), d["post_correct_scaling"]["projected"]["C5"]["shard_ms"], flush=True)
except Exception as e: print(n, "failed", repr(e)[:100])
PY
}
run bare --no-live-pmc --no-live-valu --no-clock --no-side-runs
run with_clock --no-live-pmc --no-live-valu --no-side-runs
run with_side_runs --no-live-pmc --no-live-valu --no-clock
run with_live_pmc --no-live-valu --no-clock --no-side-runs
run with_live_valu --no-live-pmc --no-clock --no-side-runs
