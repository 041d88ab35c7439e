export PM_PROF_NO_BUILD=1
mkdir -p gpurun_out/r04k
for v in head c2 head c2; do
  export PM_PROF_LIB=$PWD/protocol_amd/libpm_var_$v.so
  echo "=== $v"
  python tools/stream_prof.py 1000000 100000 | sed -n '1p;4,5p' | cut -c1-250
  python tools/stream_prof.py 100000 10000 | sed -n '1p;4,5p'
  python tools/stream_trace.py churn 2>&1 | sed -n '1p'
done
export PM_PROF_LIB=$PWD/protocol_amd/libpm_var_c2.so
python tools/stream_trace.py 100000 10000 2>&1 | sed -n '17,32p'
unset PM_PROF_LIB
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
