mkdir -p gpurun_out/r04g
timeout 400 python bench.py > gpurun_out/r04g/bench.json 2> gpurun_out/r04g/bench.err; echo rc=$?
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04g/bench.json"))
print({k:d[k] for k in ("value","ms_per_step")})
c=d["configs2"]; print("cfg2", c["ms_per_match"], c["carve_ms"])
ch=d["churn"]; print("churn", ch["ms_per_tick"], ch["split_ms_p50"])
print(d.get("pools_on_one_gpu"))
PY
