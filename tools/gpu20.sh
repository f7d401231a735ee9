cd $GRAFT_REPO_ROOT
for nt in 256 384 512 768; do
 for b in 28 mixed; do
  echo "threads2 $nt border $b: $(SID_PM_THREADS2=$nt python3 bench.py --border $b --steps 20 --warmup 3 --no-cpu-baseline --check 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"
 done
done
