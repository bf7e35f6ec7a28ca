"""-m gpu: bench.py prints ONE JSON line as its last stdout line with the fields the driver reads."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_fields(gpu_lib):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                          "--views", "12", "--segs", "60", "--neighbors", "5", "--strong-leg", "on", "--sustain-s", "0.2",
                          "--stream-leg", "on"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=ROOT,
                         env=dict(os.environ, LT_BENCH_STRONG_SCENE="30,80,6", LT_BENCH_STREAM_SCENE="40,60,6,9"))
    assert res.returncode == 0, res.stderr[-2000:]
    line = res.stdout.strip().splitlines()[-1]
    d = json.loads(line)
    # ... and the streamed leg (BASELINE config 5's stand-in; a small scene here): chunks with their neighbour closure
    s5 = d["streamed_config5"]
    assert "error" not in s5 and s5["note"] is None, s5
    assert s5["chunks"] == 5 and s5["chunk_images"] == 9 and 9 < s5["closure_images_max"] < 40 and s5["tracks"] > 0
    assert s5["candidates"] > 0 and s5["images_per_s"] > 0 and s5["device_ms_per_chunk"] > 0
    for k in ("k_gates", "k_tri_rows", "k_score3"):
        r5 = s5["roofline"][k]
        assert r5["bound"] == "hbm" and r5["achieved"] > 0 and abs(r5["frac"] - r5["achieved"] / r5["peak"]) < 1e-12
    # one run prints the weak figure and the strong-scaling leg (BASELINE config 3; a small scene here), and a sustained one
    sc3 = d["strong_config3"]
    assert "error" not in sc3, sc3
    assert sc3["scaling"] == "strong" and sc3["n_gpus"] == 1 and sc3["ms_per_step"] > 0 and sc3["value"] > 0
    assert sc3["step_with_merge_and_tail_ms"] > sc3["ms_per_step"] and len(sc3["ms_per_step_per_rank"]) == 1
    assert sc3["tracks_rank0"] > 0
    assert sc3["step_with_merge_and_tail_overlapped_ms"] > 0, sc3
    assert d["sustained_ms_per_step"] > 0 and d["sustained"]["seconds"] >= 0.15
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    # the oracle's counts for the same scene agree with the device's
    assert d["cpu_parity"]["tracks_cpu"] == d["cpu_parity"]["tracks_gpu"]
    assert d["cpu_parity"]["candidates_cpu"] == d["cpu_parity"]["candidates_gpu"]


def test_bench_collective_path_one_rank(gpu_lib):
    """The N > 1 code path of bench.py (RCCL all-gather per step, overlapped with the kernels, max over ranks) on
    the one GPU a test box has: LT_BENCH_FORCE_DIST=1 under torch.distributed.run with one rank."""
    env = dict(os.environ, LT_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"),
                          "--gpus", "1", "--steps", "3", "--warmup", "1", "--views", "12", "--segs", "60",
                          "--neighbors", "5", "--no-cpu-baseline", "--strong-leg", "on", "--sustain-s", "0.2"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, cwd=ROOT,
                         env=dict(env, LT_BENCH_STRONG_SCENE="30,80,6"))
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads(res.stdout.strip().splitlines()[-1])
    assert "error" not in d["strong_config3"] and d["strong_config3"]["ms_per_step"] > 0
    assert d["sustained_ms_per_step"] > 0
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["counts"]["candidates"] > 0 and d["tracks_whole_scene"] > 0


def test_bench_two_ranks_on_one_gpu(gpu_lib):
    """bench.py as the driver launches it for N = 2 -- torch.distributed.run, two ranks -- on the ONE GPU a test box has
    (LT_BENCH_ONE_GPU=1: both ranks on cuda:0, gloo instead of RCCL, the scene gathered through host tensors): the N > 1
    control flow of the file -- connection-weighted shards, the merge of the shards on rank 0 in its one-collective form,
    the reductions over ranks, the strong-scaling leg with its sequential and its overlapped tail -- runs to the JSON
    line, and the whole-scene track count is that of a one-rank run of the same scene."""
    common = ["--steps", "3", "--warmup", "1", "--views", "12", "--segs", "60", "--neighbors", "5", "--no-cpu-baseline",
              "--no-extras", "--strong-leg", "on", "--sustain-s", "0.2", "--scaling", "strong", "--stream-leg", "on"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LT_BENCH_STRONG_SCENE="30,80,6", LT_BENCH_STREAM_SCENE="40,60,6,9")
    res2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", "29537", os.path.join(ROOT, "bench.py"),
                           "--gpus", "2"] + common,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT,
                          env=dict(env, LT_BENCH_ONE_GPU="1"))
    assert res2.returncode == 0, res2.stderr[-3000:]
    d2 = json.loads(res2.stdout.strip().splitlines()[-1])
    res1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT, env=env)
    assert res1.returncode == 0, res1.stderr[-3000:]
    d1 = json.loads(res1.stdout.strip().splitlines()[-1])
    assert d2["n_gpus"] == 2 and d2["ranks"]["world_size"] == 2 and d2["ranks"]["backend"] == "gloo"
    assert d2["merge_note"] is None, d2["merge_note"]
    assert d2["counts"]["candidates"] == d1["counts"]["candidates"] > 0
    assert d2["tracks_whole_scene"] == d1["tracks_whole_scene"] > 0
    s2, s1 = d2["strong_config3"], d1["strong_config3"]
    assert "error" not in s2 and s2["note"] is None and s2["overlapped_note"].startswith("the host half"), s2
    assert s2["n_gpus"] == 2 and len(s2["ms_per_step_per_rank"]) == 2
    assert s2["candidates"] == s1["candidates"] > 0 and s2["tracks_rank0"] == s1["tracks_rank0"] > 0
    assert s2["step_with_merge_and_tail_ms"] > 0 and s2["step_with_merge_and_tail_overlapped_ms"] > 0
    # the streamed leg: chunks dealt round-robin to the two ranks, one gather to rank 0, the same model as one rank's
    t2, t1 = d2["streamed_config5"], d1["streamed_config5"]
    assert "error" not in t2 and t2["note"] is None and "error" not in t1 and t1["note"] is None, (t2, t1)
    assert t2["n_gpus"] == 2 and t2["chunks"] == t1["chunks"] == 5
    assert t2["candidates"] == t1["candidates"] > 0 and t2["connections"] == t1["connections"] and t2["tracks"] == t1["tracks"] > 0
