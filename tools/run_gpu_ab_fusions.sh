run() { env "$@" python bench.py --no-cpu-baseline --no-f32-mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', d['value'], d['ms_per_step'], d['roofline']['frac'])"; }
run A=0
run SAT_FUSE_RESIDUAL=1
run SAT_FUSE_BN1=1
run SAT_SHARDED_BN_MAX_TILES=1600
run A=0
run SAT_FUSE_RESIDUAL=1
