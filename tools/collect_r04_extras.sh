#!/bin/bash
# What profiles/r04_* holds beyond tools/collect_profiles.py's set, as it was collected (run on the GPU box from the
# repository root, after `python tools/collect_profiles.py r04`; outputs under gpurun_out/r04/):
#   r04_stream_anatomy_{10k,100k}.txt, r04_stream_timeline_{10k,100k,churn}.txt   PM_CARVE_PROF build (tools/stream_prof.py builds it
#                                                                                 unless PM_PROF_NO_BUILD is set)
#   r04_losing_batch_pipeline_{,cfg2_}line.json                                   round 3's carve (carve_variant 3) on the same box
#   r04_bench_n4_gloo.json                                                        four pools in four processes on the one GPU
set -u
mkdir -p gpurun_out/r04
python tools/stream_prof.py 100000 10000 > gpurun_out/r04/r04_stream_anatomy_10k.txt 2>&1
PM_PROF_NO_BUILD=1 python tools/stream_prof.py 1000000 100000 > gpurun_out/r04/r04_stream_anatomy_100k.txt 2>&1
PM_PROF_NO_BUILD=1 python tools/stream_trace.py 100000 10000 > gpurun_out/r04/r04_stream_timeline_10k.txt 2>&1
PM_PROF_NO_BUILD=1 python tools/stream_trace.py 1000000 100000 > gpurun_out/r04/r04_stream_timeline_100k.txt 2>&1
PM_PROF_NO_BUILD=1 python tools/stream_trace.py churn > gpurun_out/r04/r04_stream_timeline_churn.txt 2>&1
python bench.py --carve-variant 3 --no-extras --no-cpu-baseline > gpurun_out/r04/r04_losing_batch_pipeline_line.json 2>/dev/null
python bench.py --carve-variant 3 --config 2 --steps 5 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r04/r04_losing_batch_pipeline_cfg2_line.json 2>/dev/null
PM_STREAM_WGS=60 PM_BENCH_BACKEND=gloo PM_BENCH_SHARE_DEVICE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 \
  --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 4 --steps 5 --warmup 2 2> gpurun_out/r04/n4.err | grep '^{' > gpurun_out/r04/r04_bench_n4_gloo.json
python bench.py > gpurun_out/r04/r04_bench.json 2> gpurun_out/r04/bench.err
