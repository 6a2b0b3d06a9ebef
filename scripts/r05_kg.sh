# Developer tool (round 5): training-step kernels after a change - their tests, then the bench line's train object and the
# kernel stats of kbench's CF step.  usage: bash scripts/r05_kg.sh <tag>
TAG=${1:-kg}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo build failed; tail $O/build.log; exit 1; }
timeout 900 python -m pytest tests -m gpu -q -k "transr or kg_step or train or TransR or bpr or adam" 2>&1 | tail -4
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-leg > $O/line.json 2> $O/line.err; python3 -c "
import json;d=json.loads(open('$O/line.json').read().strip().splitlines()[-1]);print(json.dumps(d['train'])[:330])"
timeout 300 python scripts/kbench.py train --rounds 10 2>&1 | grep -v amdgpu.ids | tail -8
