export PM_PROF_NO_BUILD=1
run() { echo "== $*"; env "$@" python tools/stream_prof.py 100000 10000 | sed -n '1p;5p'; env "$@" python tools/stream_prof.py 1000000 100000 | sed -n '1p;5p'; }
run PM_STREAM_WGS=48
run PM_STREAM_WGS=96
run PM_STREAM_WGS=160
run PM_STREAM_WGS=250
