# Developer tool (round 5): split-cost triples of the d = 64 fused attention kernel inside the step (KGAT_FOLD_TILE_COST)
R=$GRAFT_REPO_ROOT; cd $R
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1 || { echo build failed; exit 1; }
for C in 64,38,1051 64,30,1051 64,46,1051 64,54,1051 64,38,600 64,38,1600 56,38,1051 64,38,1051; do
  KGAT_FOLD_TILE_COST=$C timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-leg --no-train-leg 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('cost $C', d['ms_per_step'], d['ms_per_step_steady_state'], 'att %.4f'%d['breakdown_ms']['att_score'])"
done
