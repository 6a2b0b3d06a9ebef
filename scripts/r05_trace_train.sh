# kernel traces of the CF and KG training steps (VERDICT r4 task 3): where the launches and the time go
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r05_trace_train}
mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kg --output-format csv -- python3 $R/scripts/kbench.py kg --rounds 20 > $O/kg.txt 2>&1
rocprofv3 --kernel-trace --stats -d $O/train --output-format csv -- python3 $R/scripts/kbench.py train --rounds 10 > $O/train.txt 2>&1
cat $O/kg.txt $O/train.txt | grep -v rocprofv3
