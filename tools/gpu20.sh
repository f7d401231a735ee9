cd $GRAFT_REPO_ROOT
for st in 0 40 80 160; do
 for b in 20 mixed; do
  echo "stagger $st border $b: $(SID_PM_STAGGER=$st python3 bench.py --border $b --steps 20 --warmup 3 --no-cpu-baseline --check 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")"
 done
done
