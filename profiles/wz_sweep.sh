for b in 32 64 128 256 512; do echo "blocks=$b"; UNET_WZ_BLOCKS=$b python profiles/bench_wgrad.py 2>/dev/null | head -6; done
