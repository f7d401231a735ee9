cd $GRAFT_REPO_ROOT
python3 -m cProfile -o /tmp/p.out bench.py --mode ftpm --steps 4 --warmup 1 --check 0 > /dev/null 2>&1
python3 - <<'PY'
import pstats
p = pstats.Stats('/tmp/p.out'); p.sort_stats('cumulative')
import io, sys
s = io.StringIO(); p.stream = s; p.print_stats(70); 
for ln in s.getvalue().splitlines():
    if any(k in ln for k in ('ncalls','pmlib','lib.py','ftlib','seaicedrift','domain','_capi','qhull','Delaunay','interpnd','orb.py','numpy','torch','synchronize','method')): print(ln[:170])
PY
