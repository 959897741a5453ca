# developer script: PMC passes on the ndim-228 LDS-resident stepper (run through gpurun)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/lds228_${TAG:-x}
mkdir -p $O
n=${MEMBERS:-65536}
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/sq1 -- python3 $R/tools/lds228_prof.py 2 $n 20 1 > $O/sq1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_SALU SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $O/sq2 -- python3 $R/tools/lds228_prof.py 2 $n 20 1 > $O/sq2.log 2>&1
timeout 300 rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_DCACHE_BUSY_CYCLES SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d $O/sq3 -- python3 $R/tools/lds228_prof.py 2 $n 20 1 > $O/sq3.log 2>&1
python3 - <<PY
import csv,glob,collections
for d in ('sq1','sq2','sq3'):
    fs=glob.glob('$O/'+d+'/*/*counter_collection.csv')
    if not fs: print(d,'no output'); continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'rklds' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(d, {k:'%.4g'%(sum(v)/len(v)) for k,v in acc.items()})
PY
