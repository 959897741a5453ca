# developer script: instruction-cache / scalar-cache counters of the ndim-228 LDS-resident stepper (run through gpurun)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/lds228_icache
mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o "SQC\?_[A-Z_0-9]*\(ICACHE\|IFETCH\|INST_LEVEL\|TC_\)[A-Z_0-9]*" | sort -u > $O/counters_available.txt
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/sq_counters.txt
n=${MEMBERS:-65536}
export QGS_HIP_LDS_PIPE=${PIPE:-0} QGS_HIP_LDS_MERGE=${MERGE:-0}
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/p1 -- python3 $R/tools/lds228_prof.py 2 $n 20 1 > $O/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQC_TC_REQ SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES SQC_DCACHE_BUSY_CYCLES SQ_BUSY_CYCLES SQ_IFETCH_LEVEL --kernel-trace --output-format csv -d $O/p2 -- python3 $R/tools/lds228_prof.py 2 $n 20 1 > $O/p2.log 2>&1
python3 - <<PY
import csv,glob,collections
for d in ('p1','p2'):
    fs=glob.glob('$O/'+d+'/*/*counter_collection.csv')
    if not fs: print(d,'no output'); print(open('$O/'+d+'.log').read()[-1500:]); continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'rklds' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(d, {k:'%.4g'%(sum(v)/len(v)) for k,v in acc.items()})
PY
cat $O/counters_available.txt | tr '\n' ' '
