# A/B of one environment setting on one box: ab_envval.sh VAR VALUE   (runs bench.py with VAR unset / VAR=VALUE, twice each, interleaved)
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export $1=$2; else unset $1; fi
  python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernels --no-profile --batch 0 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', '$v' == '1' and '$2' or 'unset', round(d['ms_per_step'],4), d['last_loss'])"
done
