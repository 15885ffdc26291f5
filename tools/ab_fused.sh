for d in 0 64 128 256 512; do
  MOFA_DEPHASE=$d python bench.py --arch 8 256 8 256 --steps 3 --warmup 1 --cpu-rays 0 --parity-rays 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('dephase', $d, j['value'], j['roofline']['achieved'], j['roofline']['frac'])"
done
