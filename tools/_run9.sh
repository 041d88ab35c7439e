export PM_PROF_NO_BUILD=1
mkdir -p gpurun_out/r04i
for v in base at64 at32 at16 prio; do
  export PM_PROF_LIB=$PWD/protocol_amd/libpm_var_$v.so
  echo "=== $v"
  python tools/stream_prof.py 100000 10000 | sed -n '1,2p;5p'
  python tools/stream_prof.py 1000000 100000 | sed -n '1,2p;5p'
done
export PM_PROF_LIB=$PWD/protocol_amd/libpm_var_base.so
python tools/stream_trace.py 100000 10000 > gpurun_out/r04i/tl_base.txt 2>&1; tail -16 gpurun_out/r04i/tl_base.txt
export PM_PROF_LIB=$PWD/protocol_amd/libpm_var_at32.so
python tools/stream_trace.py 100000 10000 --dump gpurun_out/r04i/trace_at32.txt > gpurun_out/r04i/tl_at32.txt 2>&1; cat gpurun_out/r04i/tl_at32.txt
python tools/stream_probe.py cfg1 0; 
