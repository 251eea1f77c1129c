for s in 1 4; do
echo "== PERM streams=$s"; python bench.py --steps 6 --warmup 2 --streams $s --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v for k,v in d['kernel_ms_per_step'].items() if k.startswith('dp_')})"
echo "== LDS streams=$s"; IPX_NO_PERM_PROFILE=1 python bench.py --steps 6 --warmup 2 --streams $s --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v for k,v in d['kernel_ms_per_step'].items() if k.startswith('dp_')})"
done
