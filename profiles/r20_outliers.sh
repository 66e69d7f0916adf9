# repeats the step timing with and without the deep-level kernels: are multi-ms outliers tied to them?
for i in 1 2 3 4 5 6 7 8; do
  for v in deep halo; do
    if [ $v = deep ]; then unset UNET_NO_DEEP_KERNELS; else export UNET_NO_DEEP_KERNELS=1; fi
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-cpp-host --no-profile --no-kernels --batch 0 2>/dev/null | python3 -c "import json,sys; print('$v', json.loads(sys.stdin.read())['ms_per_step'])"
  done
done
