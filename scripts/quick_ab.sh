# Developer tool: bench A/B of an environment switch.  usage: quick_ab.sh <tag> <ENVVAR=1> [more bench args]
TAG=$1; SW=$2; shift 2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for i in 1 2 3; do
  python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-hbm-leg "$@" > $O/a_$i.json 2>/dev/null
  env $SW python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-hbm-leg "$@" > $O/b_$i.json 2>/dev/null
done
python3 - <<PY
import json
for k in ("a","b"):
    for i in (1,2,3):
        try:
            d=json.load(open("$O/%s_%d.json"%(k,i)))
            print(k, i, d["ms_per_step"], {x: round(v*1e3,1) for x,v in d["breakdown_ms"].items()})
        except Exception as e: print(k,i,"failed",e)
PY
