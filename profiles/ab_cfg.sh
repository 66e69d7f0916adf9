# A/B of environment configurations on one box, interleaved twice:  ab_cfg.sh "A=1 B=2" "C=3" ...   ("-" = defaults)
for rep in 1 2; do
  for cfg in "$@"; do
    if [ "$cfg" = "-" ]; then e=""; else e="$cfg"; fi
    env $e python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernels --no-profile --batch 0 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-60s' % '$cfg', round(d['ms_per_step'],4), d['last_loss'])"
  done
done
