run() { env $1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernels --batch 0 --no-cpp-host --no-profile 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4))"; }
for rep in 1 2 3; do for e in "X=1" "UNET_X_HF=1" "UNET_X_HB=1" "UNET_X_HF=1,UNET_X_HB=1"; do run "$(echo $e | tr ',' ' ')"; done; done
