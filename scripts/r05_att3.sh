# Developer tool (round 5): attention kernels after a change - their tests, then the bench lines that show them.
# usage: bash scripts/r05_att3.sh <tag>
TAG=${1:-att3}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo build failed; tail $O/build.log; exit 1; }
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "att" -s > $O/pytest_att.log 2>&1; echo "pytest rc $?" >> $O/pytest_att.log
grep -n "att products\|passed\|failed\|rc " $O/pytest_att.log | tail -70
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-leg --no-train-leg"
$B > $O/d64.json 2> $O/d64.err
KGAT_ATT_TILES32=1 $B > $O/d64_tiles32.json 2> $O/d64_tiles32.err
$B --dim 128 > $O/d128.json 2> $O/d128.err

python3 - <<PY
import json
for f in ("d64","d64_tiles32","d128"):
    try:
        d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d.get("ms_per_step_steady_state"), d["breakdown_ms"]["att_score"], d.get("max_rel_err_vs_cpu"))
    except Exception as e: print(f, "ERR", e)
PY
