mkdir -p gpurun_out/r04
python tools/collect_profiles.py r04 2>&1 | tail -40
export PM_PROF_NO_BUILD=1
python tools/stream_prof.py 100000 10000 > gpurun_out/r04/r04_stream_anatomy_10k.txt 2>&1
python tools/stream_prof.py 1000000 100000 > gpurun_out/r04/r04_stream_anatomy_100k.txt 2>&1
python tools/stream_trace.py 100000 10000 > gpurun_out/r04/r04_stream_timeline_10k.txt 2>&1
python tools/stream_trace.py 1000000 100000 > gpurun_out/r04/r04_stream_timeline_100k.txt 2>&1
python tools/stream_trace.py churn > gpurun_out/r04/r04_stream_timeline_churn.txt 2>&1
# the batch pipeline (round 3's default, carve_variant 3) on the same box, for the comparison
python bench.py --carve-variant 3 --no-extras --no-cpu-baseline > gpurun_out/r04/r04_losing_batch_pipeline_line.json 2>/dev/null
python bench.py --carve-variant 3 --config 2 --steps 5 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r04/r04_losing_batch_pipeline_cfg2_line.json 2>/dev/null
# four pools in four processes on the one GPU (gloo for the bookkeeping collectives, every rank on device 0)
PM_STREAM_WGS=60 PM_BENCH_BACKEND=gloo PM_BENCH_SHARE_DEVICE=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 4 --steps 5 --warmup 2 > gpurun_out/r04/r04_bench_n4_gloo.json 2> gpurun_out/r04/n4.err
tail -c 600 gpurun_out/r04/r04_bench_n4_gloo.json
ls -la gpurun_out/r04
