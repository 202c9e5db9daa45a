cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
o=gpurun_out/r04m; mkdir -p $o
bash tools/abmed.sh 4 "--no-extras --no-steady" build/libsvx_r16.so build/libsvx_r48.so build/libsvx_r8.so > $o/ab_reads.txt 2>&1
cat $o/ab_reads.txt
