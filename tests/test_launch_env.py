"""Which device a process ends up on: svim_asm_amd/_warm.py narrows the visible devices BEFORE the HIP runtime
starts (one process per GPU; under a launcher LOCAL_RANK picks the device), bin/svim-asm derives the device from the
command line the way the real parser will read it.  Pure host logic: every case runs in a fresh interpreter with its
own environment and never loads the HIP runtime."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_py(code, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES",
                                                          "WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env or {})
    e["PYTHONPATH"] = ROOT
    out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    return json.loads(out.stdout.strip().splitlines()[-1])


RANK_CODE = """
import json, os
from svim_asm_amd import _warm
d = _warm.restrict_to_local_rank()
print(json.dumps({"device": d, "hip": os.environ.get("HIP_VISIBLE_DEVICES")}))
"""


@pytest.mark.parametrize("local", range(8))
def test_eight_ranks_each_see_their_own_device(local):
    got = run_py(RANK_CODE, {"WORLD_SIZE": "8", "RANK": str(local), "LOCAL_RANK": str(local)})
    assert got == {"device": 0, "hip": str(local)}  # the process sees ONE device, which then is device 0


@pytest.mark.parametrize("local", range(8))
def test_eight_ranks_compose_with_a_list_the_caller_set(local):
    """A job scheduler hands the launcher a permuted list: LOCAL_RANK then indexes what the caller left visible, the
    list itself is not touched (the HIP runtime resolves the index against it)."""
    listed = "7,6,5,4,3,2,1,0"
    got = run_py(RANK_CODE, {"WORLD_SIZE": "8", "RANK": str(local), "LOCAL_RANK": str(local), "HIP_VISIBLE_DEVICES": listed})
    assert got == {"device": local, "hip": listed}


DEVICE_CODE = """
import json, os, sys
from svim_asm_amd import _warm
d = _warm.restrict_to_device(int(sys.argv[1]))
out = {"device": d, "hip": os.environ.get("HIP_VISIBLE_DEVICES")}
try:
    out["logical"] = _warm.logical_device(int(sys.argv[2]))
except RuntimeError as e:
    out["logical"] = "refused"
print(json.dumps(out))
"""


def run_device(asked, later, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES",
                                                          "WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env or {})
    e["PYTHONPATH"] = ROOT
    out = subprocess.run([sys.executable, "-c", DEVICE_CODE, str(asked), str(later)], env=e, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_single_process_is_narrowed_to_the_device_it_names():
    assert run_device(3, 3) == {"device": 0, "hip": "3", "logical": 0}
    # a list the caller set is narrowed to its entry, HIP's other spelling of the variable included
    assert run_device(2, 2, {"HIP_VISIBLE_DEVICES": "4,5,6,7"}) == {"device": 0, "hip": "6", "logical": 0}
    assert run_device(1, 1, {"CUDA_VISIBLE_DEVICES": "4,5"}) == {"device": 0, "hip": "5", "logical": 0}
    # a single entry (or an index beyond the list) is left alone: the index keeps its meaning
    assert run_device(0, 0, {"HIP_VISIBLE_DEVICES": "5"}) == {"device": 0, "hip": "5", "logical": 0}


def test_a_process_narrowed_to_one_device_refuses_another():
    """The launcher sniffed device 0, the real parser then says 1: an error at once, not a failed context creation
    that is logged while the command leaves with status 0 and no VCF."""
    assert run_device(0, 1)["logical"] == "refused"


@pytest.mark.parametrize("argv,device", [(["--device", "1"], "1"), (["--device=1"], "1"), (["--dev", "1"], "1"),
                                         ([], "0"), (["--device", "x"], "0")])
def test_the_launcher_reads_the_device_like_the_real_parser(argv, device, tmp_path):
    """bin/svim-asm up to (not including) the import of the package's CLI: which device did it narrow the process to?"""
    src = open(os.path.join(ROOT, "bin", "svim-asm")).read().split('if not {"-h"')[0]
    script = tmp_path / "launcher_head.py"
    script.write_text(src.replace("os.path.dirname(os.path.dirname(os.path.abspath(__file__)))", repr(ROOT)) +
                      "\nimport json\nprint(json.dumps(os.environ.get('HIP_VISIBLE_DEVICES')))\n")
    e = {k: v for k, v in os.environ.items() if k not in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES",
                                                          "WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, str(script), "diploid", "wd", "a.bam", "b.bam", "ref.fa"] + argv, env=e,
                         capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    assert json.loads(out.stdout.strip().splitlines()[-1]) == device


def test_bench_reads_the_cgroup_throttle_counter():
    """bench.py reports how long the cgroup throttled the process inside the timed region: the reader returns a
    number (microseconds so far) where cpu.stat is exposed and None elsewhere, never raises."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(os.path.dirname(os.path.dirname(__file__)), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    v = bench._throttled_us()
    assert v is None or (isinstance(v, float) and v >= 0.0)
    w = bench._throttled_us()
    assert (v is None) == (w is None) and (v is None or w >= v)
