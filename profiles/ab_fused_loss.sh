for v in 0 1 0 1; do
  if [ $v = 1 ]; then export UNET_NO_FUSED_LOSS=1; else unset UNET_NO_FUSED_LOSS; fi
  python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernels --no-profile --batch 0 | python -c "import json,sys; print('no_fused_loss', '$v', round(json.loads(sys.stdin.read())['ms_per_step'],4))"
done
