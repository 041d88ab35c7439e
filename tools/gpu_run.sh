#!/usr/bin/env bash
# ONE script for everything that is run on the GPU box through gpurun (it replaces round 5's r05_*.sh / collect_r0*.sh one-offs):
#
#   gpurun --timeout 900 -- 'tools/gpu_run.sh <out-subdir> <step> [<step> ...]'
#
# Steps run in the order given; their logs go to gpurun_out/<out-subdir>/ and a short digest of each to stdout (what gpurun
# shows).  A step is a word, or word:argument[:argument]:
#
#   suite[:q16][:<pytest args>]  pytest tests -m gpu -x (q16: under PM_TEST_HW_QUEUES=16)
#   tests:<file>[,<file>]    the named test files only (tests/ is implied), e.g. tests:test_gpu_plugin_cxx.py
#   bench[:<bench.py args>]  the default bench line -> bench.json (+ one-line digest)
#   benchn:<N>[:<bench.py args>]  bench.py --gpus N as the driver launches it, but on ONE GPU: N ranks over gloo sharing device 0
#                            (PM_BENCH_BACKEND=gloo PM_BENCH_SHARE_DEVICE=1) — the plumbing of the N > 1 line, not its numbers
#   timing[:<rounds>]        cold matches of configs[1] and [2] (tools/variant_bench.py) + 8 churn ticks, <rounds> times (2)
#   variants:<a>,<b>         the same timings for prebuilt protocol_amd/variants/libpm_engine_<name>.so (tools/build_variants.py),
#                            product library first and last
#   env:"VAR=val VAR2=val"   the same timings of the product library under environment knobs
#   kstat[:<variant>,...]    rocprofv3 --kernel-trace --stats of 20 matches -> average duration of the carve's kernels
#   anatomy                  streaming carve taken apart with the prebuilt PM_CARVE_PROF library (tools/stream_prof.py), 10k / 100k
#   timeline                 product-like timelines (prebuilt PM_ROW_REC library, tools/stream_trace.py): 10k, 100k, churn
#   pipeline[:<args>]        a ticket's way through the streaming carve on one clock (prebuilt PM_ROW_REC library, tools/pipeline_probe.py)
#   host                     PM_TRACE_HOST marks of a cold match and of churn ticks + the rocprofv3 kernel timeline of a match
#   profiles:<rNN>           tools/collect_profiles.py <rNN>: kernel statistics, PMC passes, bench line -> gpurun_out/<rNN>/
#   fuzz[:<swarms>[:<seed>]] tools/parity_fuzz.py (engine against oracle on random swarms)
#   py:<script and args>     python tools/<script and args> (anything else)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$PWD
sub=${1:?usage: tools/gpu_run.sh <out-subdir> <step>...}; shift
out=$ROOT/gpurun_out/$sub
mkdir -p "$out"
export TMPDIR=/tmp
n=0

timings() {  # $1 = log, $2 = rounds of configs[1], $3 = rounds of configs[2]; PM_EXP_LIB / env come from the caller
  timeout 120 python tools/variant_bench.py 1 "$2" >> "$1" 2>&1
  timeout 120 python tools/variant_bench.py 2 "$3" >> "$1" 2>&1
  timeout 120 python tools/churn_probe.py 8 2>&1 | grep "^tick" | awk '{s+=$3; n++} END {if (n) printf "churn ticks mean %.3f ms over %d\n", s/n, n}' >> "$1"
}
digest() { grep -v "^  " "$1" | sed 's/defines .*: carve/carve/; s/, groups.*//'; }

for step in "$@"; do
  n=$((n+1))
  name=${step%%:*}; arg=""; [ "$step" != "$name" ] && arg=${step#*:}
  echo "##### [$n] $step"
  case $name in
    suite)
      q=""; [ "${arg%%:*}" = "q16" ] && { q="PM_TEST_HW_QUEUES=16"; arg=${arg#q16}; arg=${arg#:}; }
      env $q timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -rfE -x $arg > "$out/${n}_suite.log" 2>&1
      echo "suite rc=$?" | tee -a "$out/${n}_suite.log"; tail -4 "$out/${n}_suite.log" ;;
    tests)
      files=$(echo "$arg" | tr ',' '\n' | sed 's#^#tests/#' | tr '\n' ' ')
      timeout 900 python -m pytest $files -m gpu -q -p no:cacheprovider -rfE -x > "$out/${n}_tests.log" 2>&1
      echo "tests rc=$?" | tee -a "$out/${n}_tests.log"; tail -6 "$out/${n}_tests.log" ;;
    bench)
      timeout 600 python bench.py $arg > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?"
      python - "$out/bench.json" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("ms_per_step", d["ms_per_step"], "roofline", d.get("roofline"))
    for k in ("churn", "configs2", "merge", "dist"):
        if k in d:
            print(" ", k, {a: b for a, b in d[k].items() if not isinstance(b, (list, dict))})
except Exception as ex:
    print("no bench line:", ex)
PY
      ;;
    benchn)
      nr=${arg%%:*}; rest=""; [ "$arg" != "$nr" ] && rest=${arg#*:}
      PM_BENCH_BACKEND=gloo PM_BENCH_SHARE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$nr" \
        --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus "$nr" --steps 5 --warmup 2 $rest > "$out/bench_n${nr}_gloo.json" 2> "$out/bench_n${nr}.err"
      echo "benchn rc=$?"; tail -3 "$out/bench_n${nr}.err"
      python - "$out/bench_n${nr}_gloo.json" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("n_gpus", d["n_gpus"], "scaling", d["scaling"], "ms_per_step", d["ms_per_step"], "value", d["value"], "ranks_seen", d.get("ranks_seen"))
    print("  one_pool_sharded", {a: b for a, b in d["dist"]["one_pool_sharded"].items() if not isinstance(b, (str, dict))})
except Exception as ex:
    print("no bench line:", ex)
PY
      ;;
    timing)
      for r in $(seq 1 "${arg:-2}"); do timings "$out/${n}_timing.log" 20 8; done
      digest "$out/${n}_timing.log" ;;
    variants)
      for v in "" $(echo "$arg" | tr ',' ' ') ""; do
        lib=""; [ -n "$v" ] && lib="protocol_amd/variants/libpm_engine_$v.so"
        echo "=== variant '${v:-product}'" >> "$out/${n}_variants.log"
        PM_EXP_LIB=$lib timings "$out/${n}_variants.log" 20 8
      done
      digest "$out/${n}_variants.log" ;;
    env)
      for v in "" "$arg" ""; do
        echo "=== env '${v:--}'" >> "$out/${n}_env.log"
        env $v bash -c "$(declare -f timings); timings '$out/${n}_env.log' 20 8"
      done
      digest "$out/${n}_env.log" ;;
    kstat)
      for v in $(echo "$arg" | tr ',' ' ') "product"; do
        lib=""; [ "$v" != "product" ] && lib="$ROOT/protocol_amd/variants/libpm_engine_$v.so"
        d="$out/tr_$v"
        (cd /tmp && PM_EXP_LIB=$lib timeout -k 5 200 rocprofv3 --kernel-trace --stats -d "$d" -o t -- python "$ROOT/tools/variant_bench.py" 1 20 > "$out/${n}_kstat_$v.log" 2>&1)
        echo "=== $v" >> "$out/${n}_kstat.txt"
        db=$(find "$d" -name "t_results.db" | head -1)
        python tools/rocpd_summary.py "$db" "$out/${n}_kstat_$v.csv" > /dev/null 2>&1
        grep "carve_stream_kernel\|carve_finish\|elig_place\|pair_sweep" "$out/${n}_kstat_$v.csv" >> "$out/${n}_kstat.txt"
        rm -rf "$d"
      done
      cat "$out/${n}_kstat.txt" ;;
    anatomy)
      P=protocol_amd/variants/libpm_engine_prof.so   # (tools/build_variants.py prof=PM_CARVE_PROF, built where there is no GPU)
      PM_PROF_LIB=$P PM_PROF_NO_BUILD=1 timeout 200 python tools/stream_prof.py 100000 10000 > "$out/stream_anatomy_10k.txt" 2>&1
      PM_PROF_LIB=$P PM_PROF_NO_BUILD=1 timeout 200 python tools/stream_prof.py 1000000 100000 > "$out/stream_anatomy_100k.txt" 2>&1
      grep "T=\|a row\|proposer rows\|chain anatomy\|compute\|networks\|bitmap sweeps" "$out/stream_anatomy_10k.txt" "$out/stream_anatomy_100k.txt" ;;
    timeline)
      L=protocol_amd/variants/libpm_engine_rowrec.so
      PM_PROF_LIB=$L PM_PROF_NO_BUILD=1 timeout 200 python tools/stream_trace.py 100000 10000 > "$out/stream_timeline_10k.txt" 2>&1
      PM_PROF_LIB=$L PM_PROF_NO_BUILD=1 timeout 200 python tools/stream_trace.py 1000000 100000 > "$out/stream_timeline_100k.txt" 2>&1
      PM_PROF_LIB=$L PM_PROF_NO_BUILD=1 timeout 200 python tools/stream_trace.py churn > "$out/stream_timeline_churn.txt" 2>&1
      head -30 "$out/stream_timeline_10k.txt"; grep "chain waits\|totals" "$out/stream_timeline_100k.txt"; head -12 "$out/stream_timeline_churn.txt" ;;
    pipeline)
      PM_EXP_LIB=protocol_amd/variants/libpm_engine_rowrec.so timeout 300 python tools/pipeline_probe.py $arg > "$out/${n}_pipeline.txt" 2>&1
      echo "pipeline rc=$?"; head -150 "$out/${n}_pipeline.txt" | cut -c1-200 ;;
    host)
      PM_TRACE_HOST=1 timeout 120 python tools/host_trace.py 1 2>&1 | tail -16 > "$out/host_marks_match.txt"
      PM_TRACE_HOST=1 timeout 120 python tools/churn_probe.py 8 2>&1 | tail -40 > "$out/host_marks_churn.txt"
      (cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace -d "$out/trace" -o t -- python "$ROOT/tools/host_trace.py" 1 > "$out/trace.log" 2>&1)
      db=$(find "$out/trace" -name "t_results.db" | head -1)
      python tools/tick_timeline.py "$db" --all > "$out/timeline.txt" 2>&1
      rm -rf "$out/trace"
      cat "$out/host_marks_match.txt" "$out/timeline.txt"; grep "^tick\|^cold" "$out/host_marks_churn.txt" ;;
    profiles)
      timeout 1200 python tools/collect_profiles.py "$arg" > "$out/${n}_collect.log" 2>&1; echo "collect rc=$?"
      tail -3 "$out/${n}_collect.log" | cut -c1-600 ;;
    fuzz)
      swarms=${arg%%:*}; seed=""; [ "$arg" != "$swarms" ] && seed=${arg#*:}
      timeout 3000 python tools/parity_fuzz.py ${swarms:+--swarms $swarms} ${seed:+--seed $seed} > "$out/${n}_fuzz.txt" 2>&1; echo "fuzz rc=$?"
      tail -15 "$out/${n}_fuzz.txt" ;;
    py)
      timeout 1800 python tools/$arg > "$out/${n}_py.log" 2>&1; echo "py rc=$?"; tail -40 "$out/${n}_py.log" ;;
    *) echo "unknown step '$step'" ;;
  esac
done
