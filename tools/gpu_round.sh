#!/bin/bash
# Round evidence on the GPU box: rocprofv3 trace + PMC passes of the default bench, the bench lines of the other configs,
# and a 2-rank run of bench.py on the one GPU (gloo collectives; exercises the N > 1 code path).  Usage: tools/gpu_round.sh <tag>
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/round_$TAG
mkdir -p $OUT
cd $R
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --config 2 --no-cpu-baseline > $OUT/bench_config2.json 2> $OUT/bench_config2.err
python bench.py --config 3 --doppler-step 250 --no-cpu-baseline --steps 5 > $OUT/bench_config3.json 2> $OUT/bench_config3.err
python bench.py --config 4 --doppler-step 50 --no-cpu-baseline --steps 5 > $OUT/bench_config4.json 2> $OUT/bench_config4.err
GPSACQ_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 5 --warmup 2 > $OUT/bench_two_rank_gloo.json 2> $OUT/bench_two_rank_gloo.err
GPSACQ_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 \
    bench.py --gpus 2 --config 4 --doppler-step 50 --steps 3 --warmup 1 > $OUT/bench_two_rank_gloo_config4.json 2> $OUT/bench_two_rank_gloo_config4.err
bash tools/profile.sh $TAG
for f in $OUT/*.json; do echo "== $f"; cut -c1-400 $f; done
tail -3 $OUT/*.err | cut -c1-300
