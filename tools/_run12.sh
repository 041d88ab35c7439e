export PM_PROF_NO_BUILD=1
export PM_PROF_LIB=$PWD/protocol_amd/libpm_var_c3.so
python tools/stream_prof.py 100000 10000 | grep "stream_small"
python tools/stream_prof.py 1000000 100000 | grep "stream_small"
