cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04n; mkdir -p $o
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $o/pytest_gpu.txt
timeout 1500 python bench.py > $o/bench.json 2> $o/bench.err
tail -3 $o/pytest_gpu.txt; tail -3 $o/bench.err
