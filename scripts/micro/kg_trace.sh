R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kgtrace; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo build failed; exit 1; }
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/tr --output-format csv -- python3 $R/scripts/micro/kg_host_probe.py > $O/tr.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$O/tr/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "kgat::" in r["Name"]: print("%-60s calls %4s avg %7.1f us min %7.1f"%(r["Name"][:60],r["Calls"],float(r["AverageNs"])/1e3,float(r["MinNs"])/1e3))
PY
tail -5 $O/tr.log
