# Developer tool: the three bench lines of the driver's config (plain, 50 steps, under rocprofv3 kernel stats).  usage: bench_lines.sh <tag>
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-lines}; mkdir -p $O; cd $GRAFT_REPO_ROOT
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo "build failed"; exit 1; }   # never build under the profiler
python bench.py --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-hbm-leg > $O/bench_line_50steps.json 2>/dev/null
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/bench_stats --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 > $O/bench_line_profiled.json 2> $O/bench_profiled.err
find $O -name "*kernel_trace.csv" -delete
cd $GRAFT_REPO_ROOT
python3 - <<PY
import json
for f in ("bench_line","bench_line_50steps","bench_line_profiled"):
    d=json.load(open("$O/%s.json"%f)); print(f, d["ms_per_step"], d["roofline"].get("traffic"), (d.get("roofline_hbm") or {}).get("frac"), (d.get("roofline_att") or {}).get("traffic"))
PY
