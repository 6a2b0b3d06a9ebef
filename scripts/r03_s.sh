O=$GRAFT_REPO_ROOT/gpurun_out/r03_s; mkdir -p $O; cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
KGAT_BENCH_TRACE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-hbm-leg > $O/b20_$i.json 2> $O/b20_$i.err; grep trace $O/b20_$i.err | cut -c1-300
done
