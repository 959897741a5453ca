# developer script: effective shader clock (GRBM_GUI_ACTIVE / duration / 8 XCDs) and VALU activity of the ndim-228 stepper;
# arguments: generator knob settings "NAME=VALUE ..." (one run each), e.g. "QGS_HIP_LDS_CAP=18"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/lds228_clock
mkdir -p $O
export QGS_HIP_CACHE_DIR=/tmp/kc_variants; mkdir -p $QGS_HIP_CACHE_DIR
i=0
for v in "$@"; do
  i=$((i+1))
  for w in $v; do export $w; done
  python3 $R/tools/lds228_prof.py 2 65536 20 1 > $O/warm$i.log 2>&1
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/v$i -- python3 $R/tools/lds228_prof.py 2 65536 20 1 > $O/v$i.log 2>&1
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
print('%-60s %.2f ms  ' % (v, ms) + '  '.join('%s %.4g' % kv for kv in sorted(c.items())) + '  | GUI_ACTIVE/ns %.3f' % (c.get('GRBM_GUI_ACTIVE',0)/(ms*1e6)))
PY
done | tee $O/summary.txt
