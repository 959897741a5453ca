cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/doc228
mkdir -p $O
timeout 300 python3 $R/tools/measure_configs.py t228 > $O/measure_t228.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/lds228_prof.py 0 65536 100 2 > $O/stats.log 2>&1
TAG=final bash $R/tools/lds228_prof.sh > $O/pmc.txt 2>&1
tail -3 $O/measure_t228.txt; tail -1 $O/stats.log; tail -3 $O/pmc.txt
find $O/stats -name "*kernel_stats.csv" | head -2
