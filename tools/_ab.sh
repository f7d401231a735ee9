for b in 20 mixed; do
echo "== border $b"
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --check 0 --border $b | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('full', d['roofline']['kernel_ms_per_step'])"
for ab in SWEEP WINNER HESSIAN SAMPLING SUMS; do SID_PM_LIB=$PWD/tools/libsid_ab_$ab.so python bench.py --steps 10 --warmup 2 --no-cpu-baseline --check 0 --border $b | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('without $ab', d['roofline']['kernel_ms_per_step'])"; done
done
