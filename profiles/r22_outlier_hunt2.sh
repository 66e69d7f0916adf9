# 40 bench invocations with the deep-level kernels, every step event-timed: which step of an outlier run is slow?
for i in $(seq 1 40); do
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-cpp-host --no-profile --no-kernels --batch 0 --step-times 2> /tmp/err.txt | python3 -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['ms_per_step'], end='  ')"
  grep "step intervals" /tmp/err.txt
done
