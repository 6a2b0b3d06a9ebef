O=$GRAFT_REPO_ROOT/gpurun_out/r03_l; mkdir -p $O; cd $GRAFT_REPO_ROOT
for i in 1 2; do
for c in 64,38,1051 64,30,1051 64,24,800 64,30,800; do
KGAT_FOLD_TILE_COST=$c python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-hbm-leg > $O/bench_cost_${c}_$i.json 2>/dev/null
done
done
