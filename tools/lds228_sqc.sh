# developer script: SQC -> L2 request counters of the ndim-228 LDS-resident stepper under generator variants
# (instruction lines + scalar-data lines per second against the kernel duration), run through gpurun
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/lds228_sqc
mkdir -p $O
export QGS_HIP_CACHE_DIR=/tmp/kc_variants; mkdir -p $QGS_HIP_CACHE_DIR
i=0
for v in "$@"; do
  i=$((i+1))
  if [[ "$v" == *NOLSO=1* ]]; then export QGS_HIP_EXTRA_FLAGS="-Xclang -target-feature -Xclang -load-store-opt"; else unset QGS_HIP_EXTRA_FLAGS; fi
  for w in $v; do export $w; done
  python3 $R/tools/lds228_prof.py 2 65536 20 1 > $O/warm$i.log 2>&1      # compile outside the profiler
  timeout 300 rocprofv3 --pmc SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ SQC_TC_REQ SQC_ICACHE_REQ SQC_DCACHE_REQ SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/v$i -- python3 $R/tools/lds228_prof.py 2 65536 20 1 > $O/v$i.log 2>&1
  for w in $v; do unset ${w%%=*}; done
  python3 - "$v" $O/v$i <<'PY'
import csv,glob,collections,sys
v,d=sys.argv[1],sys.argv[2]
fs=glob.glob(d+'/*/*counter_collection.csv'); ks=glob.glob(d+'/*/*kernel_trace.csv')
acc=collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if 'rklds' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
dur=[(float(r['End_Timestamp'])-float(r['Start_Timestamp']))*1e-6 for r in csv.DictReader(open(ks[0])) if 'rklds' in r['Kernel_Name']]
c={k:sum(x)/len(x) for k,x in acc.items()}
ms=sum(dur)/len(dur)
lines=c.get('SQC_TC_INST_REQ',0)+c.get('SQC_TC_DATA_READ_REQ',0)
# 20 steps x 4 stages, 1024 workgroups on 128 CU pairs; clock from the wall time is unknown: report bytes per ns per pair
print('%-70s %.2f ms/20 steps  inst lines %.3e  data lines %.3e  -> %.1f KB per workgroup-stage, %.2f B/ns per CU pair;  VALU %.3e wait_any/wave_cycles %.2f'
      % (v, ms, c.get('SQC_TC_INST_REQ',0), c.get('SQC_TC_DATA_READ_REQ',0), lines*64/1024/(1024*80), lines*64/(ms*1e6)/128, c.get('SQ_INSTS_VALU',0), c.get('SQ_WAIT_ANY',0)/max(1,c.get('SQ_WAVE_CYCLES',1))))
PY
done | tee $O/summary.txt
