mkdir -p gpurun_out/r04g
timeout 500 python bench.py > gpurun_out/r04g/bench.json 2> gpurun_out/r04g/bench.err; echo rc=$?
tail -3 gpurun_out/r04g/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04g/bench.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["phase_ms_p50"])
print("roofline", {k:d["roofline"][k] for k in ("kernel","achieved","frac","traffic")})
print("chain", d["roofline"]["chain"])
print("carve", d["kernels"]["carve"]); print("prop", {k:v for k,v in d["kernels"]["carve_propose_kernel"].items() if k in ("ms","proposals","keys","proposals_per_group","keys_per_proposal")})
c=d["configs2"]; print("cfg2", {k:c.get(k) for k in ("ms_per_match","carve_ms","carve_kernel_ms","carve_launches","sweep_ms","error")}); print(c.get("roofline"))
ch=d["churn"]; print("churn", ch.get("ms_per_tick"), ch.get("split_ms_p50"), ch.get("error"))
print("merge", d.get("merge"))
print("pools", d.get("pools_on_one_gpu",{}).get("by_k"))
print("cpu", d.get("cpu_baseline",{}).get("value"), d.get("pcie_inclusive"))
PY
