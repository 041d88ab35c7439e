timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
bash tools/_bench.sh
