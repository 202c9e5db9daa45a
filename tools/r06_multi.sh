#!/bin/bash
# round 6: (1) full-size dataset once; (2) the .bai span fixture; (3) the command's timeline on the final thread policy;
# (4) bench.py as eight ranks sharing the one device over gloo (rendezvous, rank records, collective check)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
d=$(python3 -c "
import tempfile,sys
sys.path.insert(0,'.')
from svim_asm_amd import synth_bam
from tools import e2e_bench
d=tempfile.mkdtemp(prefix='svx_ds_'); synth_bam.write_dataset(d, **e2e_bench.dataset_args(1.0)); print(d)" 2>/dev/null | tail -1)
python3 tools/dump_bai_spans.py --dataset $d --out gpurun_out/full_bai_spans.json > /dev/null
python3 tools/cli_timeline.py $d 9 > gpurun_out/r06_cli_timeline.txt 2>&1
python3 tools/cli_timeline.py $d 9 SVX_INGEST_THREADS=32 > gpurun_out/r06_cli_timeline_32_threads.txt 2>&1
python3 tools/cli_timeline.py $d 9 SVX_BAM_DEVICE_INFLATE=100 > gpurun_out/r06_cli_timeline_leg100.txt 2>&1
grep -E "wall-clock|throttling|last mark|CPU seconds" gpurun_out/r06_cli_timeline*.txt
rm -rf $d
timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --steps 5 --warmup 2 --share-device --backend gloo --no-extras --no-cpu-baseline > gpurun_out/r06_bench_n8_shared_device.json 2> gpurun_out/r06_bench_n8.err
tail -c 1500 gpurun_out/r06_bench_n8_shared_device.json; tail -3 gpurun_out/r06_bench_n8.err
