cd /tmp && export TMPDIR=/tmp
R=/root/repo
export GABO_AB_DIMS=10
for v in duow2 duow1 bwdold; do
  k=0
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_LDS"; do
    k=$((k+1))
    rm -rf /tmp/pmc_$v$k
    GABO_HIP_LIB=$R/gabotorch_amd/libgabo_hip_$v.so timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_$v$k -o out -- python $R/tools/ab_backward.py prof > /tmp/pmc_$v$k.log 2>&1
    python3 - <<PY
import csv,glob,collections
f=glob.glob('/tmp/pmc_$v$k/**/out_counter_collection.csv',recursive=True)
if not f: print('$v no counters', open('/tmp/pmc_$v$k.log').read()[-600:]); raise SystemExit
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    kn=r['Kernel_Name']
    if 'backward' not in kn: continue
    agg[kn.split('(')[0][-40:]][r['Counter_Name']].append(float(r['Counter_Value']))
for kn,v in agg.items(): print('$v',kn,{a:(sum(b)/len(b)) for a,b in v.items()}, 'launches', len(list(v.values())[0]))
PY
  done
done
