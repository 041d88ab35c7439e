export PM_PROF_NO_BUILD=1
export PM_PROF_LIB=$PWD/protocol_amd/libpm_var_new.so
for pf in 1024 2048 4096; do
  echo "=== PF $pf"
  PM_PRUNE_FACTOR=$pf python tools/stream_prof.py 1000000 100000 | sed -n '1,2p;5p;7p'
  PM_PRUNE_FACTOR=$pf python tools/stream_prof.py 100000 10000 | sed -n '1p;5p'
done
for w in 150 210; do
  echo "=== WGS $w"
  PM_STREAM_WGS=$w python tools/stream_prof.py 1000000 100000 | sed -n '1p;5p;7p'
done
echo "=== WGS 180 PF 2048"
PM_STREAM_WGS=180 PM_PRUNE_FACTOR=2048 python tools/stream_prof.py 1000000 100000 | sed -n '1,2p;5p;7p'
