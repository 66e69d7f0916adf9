# batch-8 pass of bench.py under environment configurations:  ab_batch.sh "A=1" "B=2" ...   ("-" = defaults); prints ms per sample
for cfg in "$@"; do
  if [ "$cfg" = "-" ]; then e=""; else e="$cfg"; fi
  env $e python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernels --no-profile --batch 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); b=d.get('batch8') or {}; print('%-70s' % '$cfg', round(d['ms_per_step'],4), 'batch8 ms/sample', round(b.get('ms_per_sample',0),4), 'in_flight', b.get('in_flight'))"
done
