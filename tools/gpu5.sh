R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2e; mkdir -p $OUT
cd $R
for v in v0 v1 v5 v6; do
  echo "== $v" >> $OUT/race.txt
  SID_PM_LIB=$R/tools/ab/lib_$v.so timeout 300 python3 tools/soak_debug.py 20 2>&1 | tail -1 >> $OUT/race.txt
done
for pad in 1280 2560; do
  echo "== v0 pad $pad (3 per CU holds up to 53760 B)" >> $OUT/race.txt
  SID_PM_LDS_PAD=$pad SID_PM_LIB=$R/tools/ab/lib_v0.so timeout 300 python3 tools/soak_debug.py 20 2>&1 | tail -1 >> $OUT/race.txt
done
cat $OUT/race.txt
bash tools/ablation_run.sh r2e 20
python3 - <<'PY'
import collections
f='gpurun_out/r2e/pmc_b20.txt'
cur=None; data=collections.OrderedDict()
for ln in open(f):
    if ln.startswith('== '): cur=ln.split()[1]; data[cur]=collections.defaultdict(float); continue
    p=ln.split()
    if len(p)==6 and p[0].startswith('SQ_'):
        try: data[cur][p[0]]+=float(p[4])
        except: pass
keys=sorted({k for d in data.values() for k in d})
print('%-28s'%'counter'+''.join('%14s'%v for v in data))
for k in keys: print('%-28s'%k+''.join('%14.4g'%data[v][k] for v in data))
PY
