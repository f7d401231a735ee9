cd $GRAFT_REPO_ROOT
for np in 0 1; do
  if [ $np = 1 ]; then export SID_PM_NO_PAIRED=1; else unset SID_PM_NO_PAIRED; fi
  for cfg in "--angles 3" "--angles 3 --border 20" "--angles 1" "--angles 1 --border 20" "--angles 3 --border 50"; do echo "no_paired=$np $cfg: $(python3 bench.py $cfg --steps 20 --warmup 3 --no-cpu-baseline --check 64 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['parity_check']['ok'])")"; done
done
