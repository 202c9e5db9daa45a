cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
SVX_BAM_DEBUG=1 timeout 900 python3 tools/e2e_bench.py --scale 0.25 --repeat 1 --ranks "" 2>&1 >/dev/null | grep -E "seq_slices:|svx_bam_load" | tail -8
