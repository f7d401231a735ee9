cd $GRAFT_REPO_ROOT
for sw in "20 3" "20 40" "200 100"; do set -- $sw; for b in 20 mixed; do python3 bench.py --border $b --steps $1 --warmup $2 --no-cpu-baseline --check 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps/warmup $sw border $b:', d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"; done; done
