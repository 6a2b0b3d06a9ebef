# Developer tool (round 5): split-cost triples of the d = 128 fused attention kernel inside the step (KGAT_FOLD_TILE_COST)
R=$GRAFT_REPO_ROOT; cd $R
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1 || { echo build failed; exit 1; }
for C in 64,12,700 64,18,700 64,24,700 64,32,700 64,24,1200 64,24,400 64,40,1000; do
  KGAT_FOLD_TILE_COST=$C timeout 300 python bench.py --dim 128 --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-leg --no-train-leg 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('cost $C', d['ms_per_step'], d['ms_per_step_steady_state'], 'att %.4f'%d['breakdown_ms']['att_score'])"
done
