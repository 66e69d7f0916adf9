for b in 128 256 512 1024 2048; do echo "blocks=$b"; UNET_WZ_BLOCKS=$b python profiles/bench_wgrad.py 2>/dev/null | head -5; done
