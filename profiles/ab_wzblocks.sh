for v in 0 256 0 256; do
  if [ $v = 0 ]; then unset UNET_WZ_BLOCKS; else export UNET_WZ_BLOCKS=$v; fi
  python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-kernels --no-profile --batch 0 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('UNET_WZ_BLOCKS', '$v', round(d['ms_per_step'],4))"
done
