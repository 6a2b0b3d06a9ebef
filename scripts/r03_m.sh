O=$GRAFT_REPO_ROOT/gpurun_out/r03_m; mkdir -p $O; cd $GRAFT_REPO_ROOT
for i in 1 2; do
for c in 64,38,1051 64,46,1051 64,54,1051 64,38,1400 64,46,1400 64,64,1051; do
KGAT_FOLD_TILE_COST=$c python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-hbm-leg > $O/bench_cost_${c}_$i.json 2>/dev/null
done
done
