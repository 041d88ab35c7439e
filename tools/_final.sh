timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python tools/collect_profiles.py r04 --skip-extra-pmc 2>&1 | tail -3
bash tools/collect_r04_extras.sh
ls gpurun_out/r04 | head -40
