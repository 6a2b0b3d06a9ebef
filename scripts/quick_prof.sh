# Developer tool: the GPU suite, then the driver's bench command under rocprofv3 kernel stats.  usage: quick_prof.sh <tag> [notests]
TAG=${1:-q}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo "build failed"; exit 1; }   # never build under the profiler
if [ "$2" != "notests" ]; then
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
tail -4 $O/pytest_gpu.log
fi
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/bench_stats --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-leg > $O/bench_line_profiled.json 2> $O/bench_profiled.err
find $O -name "*kernel_trace.csv" -delete
find $O -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -d, -f1-4 {} | head -24 | cut -c1-150'
python3 -c "
import json,sys
d=json.load(open('$O/bench_line_profiled.json'))
print('ms_per_step', d['ms_per_step'], json.dumps(d['breakdown_ms']))
"
