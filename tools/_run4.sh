mkdir -p gpurun_out/r04d; export PM_PROF_NO_BUILD=1
timeout 900 python tools/stream_probe.py all > gpurun_out/r04d/probe.txt 2>&1; echo probe rc=$?
python - <<'PY'
import json
for l in open("gpurun_out/r04d/probe.txt"):
    try: d=json.loads(l)
    except Exception: print(l.strip()[:300]); continue
    i=d.get("info",{})
    print(d.get("case"), "match", d.get("match"), "ms_warm", d.get("ms_warm"), "tickets", i.get("stream_tickets"), "timeouts", i.get("stream_timeouts"), "pre_used", i.get("stream_pre_used"), "pre_lost", i.get("stream_pre_lost"), "slow", i.get("slow_steps"), "fallbacks", i.get("prune_fallbacks"), d.get("first_diff"))
PY
( timeout 120 python tools/stream_prof.py 100000 10000; timeout 200 python tools/stream_prof.py 1000000 100000 ) > gpurun_out/r04d/prof.txt 2>&1
cat gpurun_out/r04d/prof.txt
