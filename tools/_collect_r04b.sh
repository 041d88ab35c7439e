mkdir -p gpurun_out/r04
export PM_PROF_NO_BUILD=1
python tools/stream_prof.py 100000 10000 > gpurun_out/r04/r04_stream_anatomy_10k.txt 2>&1
python tools/stream_prof.py 1000000 100000 > gpurun_out/r04/r04_stream_anatomy_100k.txt 2>&1
python tools/stream_trace.py 100000 10000 > gpurun_out/r04/r04_stream_timeline_10k.txt 2>&1
python tools/stream_trace.py 1000000 100000 > gpurun_out/r04/r04_stream_timeline_100k.txt 2>&1
python tools/stream_trace.py churn > gpurun_out/r04/r04_stream_timeline_churn.txt 2>&1
timeout 500 python bench.py > gpurun_out/r04/r04_bench.json 2> gpurun_out/r04/bench.err
head -3 gpurun_out/r04/r04_stream_anatomy_10k.txt; head -2 gpurun_out/r04/r04_stream_timeline_churn.txt
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04/r04_bench.json"))
print(d["ms_per_step"], d["roofline"]["traffic"], d["roofline"]["traffic_source"][:30], d["configs2"]["roofline"]["traffic"], d["churn"]["carve_hbm_bytes_per_tick"], d["churn"]["ms_per_tick"], d["configs2"]["ms_per_match"])
PY
