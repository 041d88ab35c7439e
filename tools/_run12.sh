export PM_PROF_NO_BUILD=1
mkdir -p gpurun_out/r04k
for v in c3 pa c3 pa; do
  export PM_PROF_LIB=$PWD/protocol_amd/libpm_var_$v.so
  echo "=== $v"
  python tools/stream_prof.py 1000000 100000 | sed -n '1p'
  python tools/stream_prof.py 100000 10000 | sed -n '1p'
  python tools/stream_trace.py churn 2>&1 | sed -n '1p'
done
export PM_PROF_LIB=$PWD/protocol_amd/libpm_var_pa.so
python tools/stream_trace.py churn 2>&1 | sed -n '2,16p'
python tools/stream_trace.py 100000 10000 2>&1 | sed -n '2,16p'
unset PM_PROF_LIB
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
