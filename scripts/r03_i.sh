O=$GRAFT_REPO_ROOT/gpurun_out/r03_i; mkdir -p $O; cd $GRAFT_REPO_ROOT
python scripts/micro/att_variants_ab.py --dim 64 --rounds 12 --caps 256 --costs 64,38,1051:64,30,1051:64,46,1051:64,38,700:64,38,1400:64,24,800:64,30,1400:64,20,1051 > $O/att64_costs.log 2>&1
python scripts/micro/att_variants_ab.py --dim 64 --rounds 12 --caps 128,192,320,512 --costs 64,38,1051 > $O/att64_caps.log 2>&1
python scripts/micro/att_variants_ab.py --dim 128 --rounds 8 --caps 128,512 --costs 64,12,700 > $O/att128_caps.log 2>&1
python scripts/micro/att_variants_ab.py --dim 64 --workload last-fm --rounds 8 --caps 256 --costs 64,38,1051:64,30,1051:64,38,700:64,24,800 > $O/att64_lastfm.log 2>&1
cat $O/att64_costs.log $O/att64_caps.log $O/att128_caps.log $O/att64_lastfm.log
