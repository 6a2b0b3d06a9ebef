# quick GPU check: usage  bash scripts/r05_quick.sh <tag> <pytest args...>
TAG=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 900 python -m pytest "$@" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -25 $O/pytest.log
