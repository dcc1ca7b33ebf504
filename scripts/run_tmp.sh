bash scripts/profile_round.sh r06 c > gpurun_out/profile_round.log 2>&1
BATCHES="256 4096" bash scripts/profile_train_squared.sh > gpurun_out/prof_sq.log 2>&1
bash scripts/profile_train.sh r06 c fused > gpurun_out/prof_train.log 2>&1
ls gpurun_out/r06 gpurun_out/train_sq
