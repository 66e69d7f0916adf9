# round 3, first A/B set on one box: GPU tests, then the stream-ordering knobs (fork sharing, event fence, pack split, one loss fork)
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests -m gpu -x -q > gpurun_out/r07b_tests.log 2>&1
echo "tests rc $?" >> gpurun_out/r07b_tests.log
tail -3 gpurun_out/r07b_tests.log
(
bash profiles/ab_envval.sh UNET_FORK_EVERY 1
bash profiles/ab_envval.sh UNET_FORK_EVERY 6
bash profiles/ab_envval.sh UNET_EVENT_SYSFENCE 1
bash profiles/ab_env.sh UNET_PACK_ONE_LAUNCH
bash profiles/ab_env.sh UNET_LOSS_FORK_PER_LEVEL
) > gpurun_out/r07b_ab.txt 2>&1
cat gpurun_out/r07b_ab.txt
