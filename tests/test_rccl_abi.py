"""The C-ABI companion for multi-GPU hosts, liblimap_amd_rccl.so (include/limap_amd_rccl.h): the library loads and exports
what the header declares, and its sharding rule is limap_amd.dist.shard_bounds' (CPU); -m gpu: a one-rank RCCL communicator
takes the same path as N ranks would -- the all-gather of the packed scene into the receive buffer, the context initialised
from it, the per-step refresh -- and the tracks are the oracle's.  (Two ranks need two GPUs: RCCL refuses a device twice.)"""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "limap_amd", "liblimap_amd_rccl.so")
HDR = os.path.join(ROOT, "include", "limap_amd_rccl.h")


def _load():
    from limap_amd import _capi
    _capi.load_library()  # liblimap_amd.so first (and the process's one HIP runtime, see _capi)
    return C.CDLL(LIB, mode=C.RTLD_GLOBAL)


def test_library_exports_every_symbol_of_the_header():
    if not os.path.exists(LIB):
        pytest.fail("limap_amd/liblimap_amd_rccl.so is not built (python -c 'import __graft_entry__ as g; g.build()')")
    L = _load()
    names = re.findall(r"\b(lt_dist_\w+)\s*\(", open(HDR).read())
    assert len(set(names)) >= 8
    for n in set(names):
        assert hasattr(L, n), n


def test_shard_bounds_is_the_python_rule():
    from limap_amd import dist as ltdist
    L = _load()
    L.lt_dist_shard_bounds.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    rng = np.random.default_rng(5)
    for n, world in [(1, 1), (7, 2), (100, 8), (13, 16), (1000, 8), (5, 8)]:
        for weights in (None, rng.uniform(0.0, 5.0, n), np.r_[np.zeros(n // 2), np.ones(n - n // 2)]):
            out = np.zeros(world + 1, np.int64)
            w = None if weights is None else np.ascontiguousarray(weights, np.float64)
            rc = L.lt_dist_shard_bounds(n, world, None if w is None else w.ctypes.data_as(C.POINTER(C.c_double)),
                                        out.ctypes.data_as(C.POINTER(C.c_int64)))
            assert rc == 0
            assert out.tolist() == [int(b) for b in ltdist.shard_bounds(n, world, weights)], (n, world)


_WORKER = r'''
import ctypes as C, json, os, sys
import numpy as np
ROOT = sys.argv[1]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from limap_amd import _capi, synthetic as syn
lib = _capi.load_library()
R = C.CDLL("/opt/rocm/lib/librccl.so", mode=C.RTLD_GLOBAL)
D = C.CDLL(os.path.join(ROOT, "limap_amd", "liblimap_amd_rccl.so"))
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
class UID(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]
uid = UID()
rc = R.ncclGetUniqueId(C.byref(uid))
if rc != 0:
    print(json.dumps({"skip": "ncclGetUniqueId failed with %d" % rc})); sys.exit(0)
comm = C.c_void_p()
R.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UID, C.c_int]
rc = R.ncclCommInitRank(C.byref(comm), 1, uid, 0)
if rc != 0:
    print(json.dumps({"skip": "ncclCommInitRank failed with %d" % rc})); sys.exit(0)
stream = C.c_void_p()
assert hip.hipStreamCreate(C.byref(stream)) == 0
sc = syn.make_scene(n_views=12, n_segs=70, n_neighbors=5, seed=31)
cfg = syn.default_triangulation_cfg()
ctx = _capi.Context(cfg_dict=cfg, device=0)
ctx.set_ranges(*sc.ranges)
D.lt_dist_create.restype = C.c_void_p
D.lt_dist_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
D.lt_dist_last_error.restype = C.c_char_p
D.lt_dist_last_error.argtypes = [C.c_void_p]
for f in ("lt_dist_load_local", "lt_dist_all_gather_scene", "lt_dist_merge_shards", "lt_dist_my_images", "lt_dist_destroy"):
    getattr(D, f).argtypes = None
ids = np.ascontiguousarray(sc.img_ids, np.int32); so = np.ascontiguousarray(sc.seg_off, np.int64)
w = np.ascontiguousarray([len(sc.neighbors[int(i)]) for i in sc.img_ids], np.float64)
hctx = ctx.h if isinstance(ctx.h, C.c_void_p) else C.c_void_p(ctx.h)
d = D.lt_dist_create(hctx, comm, stream, 0, 1, len(ids), ids.ctypes.data, so.ctypes.data, w.ctypes.data)
assert d, "lt_dist_create"
d = C.c_void_p(d)
def chk(rc):
    assert rc == 0, D.lt_dist_last_error(d).decode()
a, b = C.c_int(), C.c_int()
chk(D.lt_dist_my_images(d, C.byref(a), C.byref(b)))
assert (a.value, b.value) == (0, len(ids))
k = np.ascontiguousarray(sc.kvec, np.float64); q = np.ascontiguousarray(sc.qvec, np.float64)
t = np.ascontiguousarray(sc.tvec, np.float64); s = np.ascontiguousarray(sc.segs, np.float64)
P = lambda x: C.c_void_p(x.ctypes.data)
chk(D.lt_dist_load_local(d, P(k), P(q), P(t), P(s)))
chk(D.lt_dist_all_gather_scene(d))          # first gather: lt_init_device + chunks
for i in sc.img_ids:
    m = sc.matches_of(int(i)); nb = list(m.keys())
    off = np.zeros(len(nb) + 1, np.int64); off[1:] = np.cumsum([len(m[x]) for x in nb])
    ctx.triangulate_image(int(i), nb, off, np.concatenate([m[x] for x in nb], 0))
ctx.upload()
chk(D.lt_dist_load_local(d, P(k), P(q), P(t), P(s)))
chk(D.lt_dist_all_gather_scene(d))          # per-step form: lt_refresh_scene_chunks from the receive buffer
ctx.run_device()
n = C.c_int64(-1)
chk(D.lt_dist_merge_shards(d, C.c_int64(1024), C.byref(n)))
assert n.value == 0                         # one rank: nothing to merge
ctx.compute_tracks()
tr = ctx.get_tracks()
from oracle import oracle as ora
from helpers import compare_tracks, run_oracle
ora.build()
O = run_oracle(ora, sc, cfg)
compare_tracks({k_: np.asarray(v) for k_, v in tr.items()}, O.ComputeLineTracks())
st = ctx.stats()
D.lt_dist_destroy(d)
print(json.dumps({"tracks": st["tracks"], "candidates": st["candidates"]}))
'''


@pytest.mark.gpu
def test_one_rank_rccl_gather_init_refresh_and_tracks(gpu_lib):
    import json
    res = subprocess.run([sys.executable, "-c", _WORKER, ROOT], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         timeout=600, cwd=ROOT, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LIMAP_AMD_SYSTEM_HIP="1"))
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]  # (RCCL prints its banner to stdout)
    assert lines, (res.stdout[-500:], res.stderr[-1500:])
    d = json.loads(lines[-1])
    if "skip" in d:
        pytest.skip(d["skip"])
    assert d["tracks"] > 0 and d["candidates"] > 0


# ---- the plain-C host (examples/rccl_host.c -> limap_amd/rccl_host): the C-ABI boundary carries the multi-GPU path ----
HOST = os.path.join(ROOT, "limap_amd", "rccl_host")


def test_plain_c_host_is_built_against_the_two_headers_only():
    if not os.path.exists(LIB):
        pytest.skip("liblimap_amd_rccl.so is not built on this box (no RCCL): no C host either")
    assert os.path.exists(HOST), "limap_amd/rccl_host is not built (make -C limap_amd/csrc rccl)"
    src = open(os.path.join(ROOT, "examples", "rccl_host.c")).read()
    ours = [h for h in re.findall(r'#include "([^"]+)"', src)]
    assert sorted(ours) == ["limap_amd.h", "limap_amd_rccl.h"], ours  # nothing of the product's internals, no C++


def _run_c_host(tmp_path, world, n_gpus_needed):
    # (device count from a child process: importing torch HERE, after liblimap_amd_rccl.so has pulled in the system's librccl,
    # would leave this process with torch bound to a foreign RCCL -- it aborts at exit)
    n_dev = int(subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                               stdout=subprocess.PIPE, text=True, timeout=300).stdout.strip().splitlines()[-1])
    if n_dev < n_gpus_needed:
        pytest.skip(f"needs {n_gpus_needed} GPUs (RCCL refuses one device twice), this box has {n_dev}")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from write_scene_bin import fnv_members, write_scene_bin
    from limap_amd import synthetic as syn
    from helpers import run_product
    sc = syn.make_scene(n_views=24, n_segs=90, n_neighbors=6, seed=41)
    scene_bin, idf = str(tmp_path / "scene.bin"), str(tmp_path / "nccl_id")
    write_scene_bin(scene_bin, sc)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    outs = [str(tmp_path / f"out{r}.txt") for r in range(world)]
    procs = [subprocess.Popen([HOST, str(r), str(world), scene_bin, idf, outs[r]], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
    res = [p.communicate(timeout=600) for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, (r, res[r][1][-2000:])
    line = open(outs[0]).read().split()
    got = dict(zip(line[0::2], line[1::2]))
    # the one-process result through the Python mirror (itself held to the oracle by the parity tests)
    T = run_product(sc, {})  # the C host runs lt_config_default: the reference's class defaults, no yaml on top
    T.ComputeLineTracks()
    tr = T.context().get_tracks()
    assert int(got["tracks"]) == len(tr["off"]) - 1 > 0 and int(got["members"]) == len(tr["image_ids"])
    assert int(got["fnv"], 16) == fnv_members(tr["off"], tr["image_ids"], tr["line_ids"])
    return got


@pytest.mark.gpu
def test_plain_c_host_one_rank(gpu_lib, tmp_path):
    """the whole sequence of examples/rccl_host.c with a one-rank communicator (what a one-GPU box can run)"""
    if not os.path.exists(HOST):
        pytest.skip("limap_amd/rccl_host is not built")
    _run_c_host(tmp_path, 1, 1)


@pytest.mark.gpu
def test_plain_c_host_two_ranks(gpu_lib, tmp_path):
    """two processes, two GPUs, ncclCommInitRank over a file-shared unique id: all-gather, sharded run, one-collective
    merge, ComputeLineTracks on rank 0 -- the tracks of the one-process run"""
    if not os.path.exists(HOST):
        pytest.skip("limap_amd/rccl_host is not built")
    got = _run_c_host(tmp_path, 2, 2)
    assert int(got["keys"]) > 0
