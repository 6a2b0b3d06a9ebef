#!/bin/bash
# round 6 evidence run: the driver's bench command (plain), then the same under rocprofv3 kernel stats (step only),
# and a kernel trace of the first steps (scripts/first_steps_trace.py)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd $R
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo build failed; exit 1; }
python3 bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; echo "bench rc $?"
python3 - <<PY
import json
d=json.load(open("$O/bench_line.json"))
print("ms_per_step", d["ms_per_step"], "steady", d.get("ms_per_step_steady_state"), "lazy", d.get("ms_per_step_lazy_opt_in"))
print("roofline", {k: d["roofline"][k] for k in ("achieved","peak","frac","algorithmic_over_hbm_peak","frac_of_gather_ceiling","avg_ms")})
t=d.get("train",{}); print("train", {k: t.get(k) for k in ("epoch_measured_s","kg_phase_ms_per_iteration","cf_step_ms","kg_step_ms","eval_ms","error")}); print(t.get("epoch_measured")); print(t.get("epoch_model"))
print("hbm", (d.get("roofline_hbm") or {}).get("frac"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
print("keys", list(d.keys()))
PY
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/step_stats -o step --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-leg --no-train-leg > $O/bench_line_profiled.json 2> $O/bench_profiled.err
cp $O/step_stats/step_kernel_stats.csv $O/r06_bench_kernel_stats_step_only.csv
python3 $R/scripts/first_steps_trace.py $O/step_stats/step_kernel_trace.csv > $O/r06_first_steps_trace.txt 2>&1
rm -f $O/step_stats/step_kernel_trace.csv
cut -d, -f1-4 $O/r06_bench_kernel_stats_step_only.csv | head -14 | cut -c1-150
head -60 $O/r06_first_steps_trace.txt
