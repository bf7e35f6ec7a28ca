"""-m gpu: the N > 1 path over RCCL (backend "nccl") with real processes, one per GPU: the single all-gather of the
per-image payload into device memory, every rank triangulating its connection-weighted shard, the packed tensor
gather of the per-node results to rank 0, ComputeLineTracks there -- against the oracle.

world_size 2 needs two GPUs (skipped with the reason on a 1-GPU box; the driver's multi-GPU tier runs it);
world_size 1 takes the same code path -- process group, collectives, chunk pointers -- on one GPU."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q, backend="nccl", merge="host", min_outer=0):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    # backend "gloo": every rank on cuda:0 (RCCL needs a device per rank; gloo moves host tensors, the gathered scene is
    # copied to the device afterwards) -- the two-rank shard / merge path with two real contexts on a 1-GPU box
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    cdev = dev if backend == "nccl" else torch.device("cpu")
    try:
        from limap_amd import _capi, dist as ltdist, synthetic as syn
        sc = syn.make_scene(n_views=14, n_segs=90, n_neighbors=6, seed=21)
        cfg = syn.default_triangulation_cfg(min_num_outer_edges=min_outer)
        weights = np.array([len(sc.neighbors[int(i)]) for i in sc.img_ids], float)
        mine = ltdist.shard_images(sc.img_ids, rank, world, weights)
        g = ltdist.SceneGather(sc.img_ids, sc.seg_off, rank, world, cdev, weights=weights, force_collective=True)
        # every rank loads ONLY its own slice; the rest must arrive through the collective
        a, b = g.bounds[rank], g.bounds[rank + 1]
        kv, qv, tv, sg = sc.kvec.copy(), sc.qvec.copy(), sc.tvec.copy(), sc.segs.copy()
        mask = np.ones(sc.n_images, bool); mask[a:b] = False
        kv[mask] = np.nan; qv[mask] = np.nan; tv[mask] = np.nan
        smask = np.ones(len(sg), bool); smask[sc.seg_off[a]:sc.seg_off[b]] = False
        sg[smask] = np.nan
        g.load_local(kv, qv, tv, sg)
        d_k, d_q, d_t, d_s = g.all_gather()
        ok = bool(torch.isfinite(d_s).all().item()) and np.array_equal(d_k.cpu().numpy(), sc.kvec)
        if backend != "nccl":
            d_k, d_q, d_t, d_s = d_k.to(dev), d_q.to(dev), d_t.to(dev), d_s.to(dev)
        ctx = _capi.Context(cfg_dict=cfg, device=dev.index)
        ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        ctx.set_ranges(*sc.ranges)
        ctx.init_device(sc.img_ids, d_k.data_ptr(), d_q.data_ptr(), d_t.data_ptr(), sc.seg_off, d_s.data_ptr())
        for i in mine:
            m = sc.matches_of(int(i))
            nb = list(m.keys())
            off = np.zeros(len(nb) + 1, np.int64); off[1:] = np.cumsum([len(m[k]) for k in nb])
            ctx.triangulate_image(int(i), nb, off, np.concatenate([m[k] for k in nb], 0))
        ctx.upload()
        if backend == "nccl":
            ctx.set_scene_chunks(*g.chunk_pointers())
            h = g.gather_async()
            h.wait()
            ctx.refresh_scene_chunks()   # the per-step path: invariants rebuilt straight from the receive buffer
        ctx.run_device()
        res = None
        if merge in ("device", "device1", "device1_small"):
            # round 4: nothing is read back -- keys and node slices travel device to device (host tensors under gloo),
            # rank 0's device tail runs over the whole scene.  "device1": the one-collective form (round 5: static node
            # ranges + a key capacity, the counts ride in the blob's header); "device1_small": a capacity that does not
            # hold -- the rank that cannot send and rank 0 both raise, nobody hangs
            kw = {}
            if merge != "device":
                kw["all_ranges"] = [(int(sc.seg_off[g.bounds[r]]), int(sc.seg_off[g.bounds[r + 1]])) for r in range(world)]
                kw["key_cap"] = 1 if merge == "device1_small" else 64 * 1024
            try:
                n_keys = ltdist.merge_shards_device(ctx, (int(sc.seg_off[a]), int(sc.seg_off[b])), rank, world,
                                                    dev if backend == "nccl" else None, **kw)
            except RuntimeError as e:
                q.put((rank, ok, dict(error=str(e))))
                dist.barrier()
                return
            if rank == 0:
                ctx.compute_tracks()
                t = ctx.get_tracks()
                res = dict(tracks={k: np.asarray(v) for k, v in t.items()}, best=None, imported=sc.n_images - len(mine),
                           mine=len(mine), n_keys=n_keys, valid_flags=np.asarray(ctx.get_valid_flags()))
        else:
            ctx.download()
            # device None: merge_shards_on_rank0 picks it from the process group's backend (host tensors under gloo)
            n_imp = ltdist.merge_shards_on_rank0(ctx, mine, rank, world, dev if backend == "nccl" else None)
            if rank == 0:
                ctx.compute_tracks()
                t = ctx.get_tracks()
                b_ = ctx.get_best()
                res = dict(tracks={k: np.asarray(v) for k, v in t.items()}, best={k: np.asarray(v) for k, v in b_.items()},
                           imported=n_imp, mine=len(mine))
        q.put((rank, ok, res))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _run(world, backend="nccl", merge="host", min_outer=0):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, backend, merge, min_outer)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return sorted(out, key=lambda x: x[0])


def _check(out, world, oracle, min_outer=0):
    from limap_amd import synthetic as syn
    from helpers import compare_best, compare_tracks, run_oracle
    assert all(ok for _, ok, _ in out)
    res = out[0][2]
    sc = syn.make_scene(n_views=14, n_segs=90, n_neighbors=6, seed=21)
    assert res["imported"] + res["mine"] == sc.n_images
    O = run_oracle(oracle, sc, syn.default_triangulation_cfg(min_num_outer_edges=min_outer))
    if res["best"] is not None:  # (the device merge leaves the other shards' per-node results on the device)
        compare_best(res["best"], O.get_best())
    compare_tracks(res["tracks"], O.ComputeLineTracks())


@pytest.mark.parametrize("merge", ["host", "device"])
def test_rccl_path_world1(gpu_lib, oracle, merge):
    _check(_run(1, merge=merge), 1, oracle)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the 1-GPU box runs the world_size 1 form)")
@pytest.mark.parametrize("merge", ["host", "device"])
def test_rccl_path_world2(gpu_lib, oracle, merge):
    _check(_run(2, merge=merge), 2, oracle)


def test_two_ranks_two_contexts_on_one_gpu_gloo(gpu_lib, oracle):
    """N > 1 end to end where only one GPU exists: two processes, two contexts on cuda:0, gloo process group -- shards,
    the scene all-gather, per-shard triangulation, merge_shards_on_rank0 (device from the backend) and the tail on rank 0
    against the oracle's whole-scene result."""
    _check(_run(2, "gloo"), 2, oracle)


def test_two_ranks_device_merge_on_one_gpu_gloo(gpu_lib, oracle):
    """The device-to-device merge (round 4) with two real contexts on cuda:0 over a gloo group: rank 1 builds the keys of
    its nodes on its device and exports keys + node slices (lt_shard_export, into the gather's host tensors), rank 0
    imports them into its device arrays (lt_shard_import) and runs the DEVICE tail over the whole scene: same tracks
    as the oracle's single-process run."""
    out = _run(2, "gloo", merge="device")
    _check(out, 2, oracle)
    assert out[0][2]["n_keys"] > 0



@pytest.mark.parametrize("min_outer", [1, 2])
def test_two_ranks_device_merge_with_the_node_filter(gpu_lib, oracle, min_outer):
    """min_num_outer_edges > 0 over imported shards (round 6; refused before): with the filter on a shard ships DIRECTED keys,
    rank 0 runs filterNodeByNumOuterEdges (global_line_triangulator.cc:168-232) over the merged list on its device, then the
    tail as usual -- tracks and the per-node valid flags of the oracle's single-process run."""
    from helpers import run_oracle
    from limap_amd import synthetic as syn
    out = _run(2, "gloo", merge="device1", min_outer=min_outer)
    _check(out, 2, oracle, min_outer)
    sc = syn.make_scene(n_views=14, n_segs=90, n_neighbors=6, seed=21)
    n_with = len(run_oracle(oracle, sc, syn.default_triangulation_cfg(min_num_outer_edges=min_outer)).ComputeLineTracks()["off"]) - 1
    n_without = len(run_oracle(oracle, sc, syn.default_triangulation_cfg()).ComputeLineTracks()["off"]) - 1
    assert 0 < n_with < n_without  # the filter really removes nodes, and tracks remain
    flags = out[0][2]["valid_flags"].astype(bool)
    assert 0 < flags.sum() < len(flags)


def test_two_ranks_one_collective_merge_on_one_gpu_gloo(gpu_lib, oracle):
    """The merge as ONE gather (round 5): every rank knows every node range (the sharding is deterministic), the key
    counts ride in the 64-byte header of the blob -- no size exchange.  Same tracks as the oracle's single-process run."""
    out = _run(2, "gloo", merge="device1")
    _check(out, 2, oracle)
    assert out[0][2]["n_keys"] > 0


def test_one_collective_merge_fails_loudly_when_the_keys_do_not_fit(gpu_lib):
    out = _run(2, "gloo", merge="device1_small")
    assert all("error" in r[2] for r in out), out
    assert "key_cap" in out[1][2]["error"] and "could not send" in out[0][2]["error"]


def test_shard_import_rejects_keys_outside_the_scene(gpu_lib):
    """lt_shard_import range-checks the keys it is handed (they index the per-node arrays in the similarity kernel): a
    key that names a node >= G, or is not (min << kb | max), is an argument error -- and a clean re-import still works."""
    from limap_amd import _capi, synthetic as syn, triangulation as tri
    sc = syn.make_scene(n_views=10, n_segs=80, n_neighbors=5, seed=77)
    T = tri.GlobalLineTriangulator(syn.default_triangulation_cfg())
    T.SetRanges(sc.ranges)
    T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, [sc.segs_of(j) for j in range(sc.n_images)])
    T.TriangulateAll({int(i): sc.matches_of(int(i)) for i in sc.img_ids})
    ctx = T.context()
    ctx.upload()
    ctx.run_device()
    n = ctx.shard_count()
    assert n > 0
    G = int(sc.seg_off[-1])
    nodes = np.zeros(ctx.shard_node_bytes() * G, np.uint8)
    keys = np.zeros(n, np.uint64)
    ctx.shard_build(n)                           # as a sending rank: its own keys only
    ctx.shard_export(0, G, nodes.ctypes.data, keys.ctypes.data)
    assert ctx.shard_count() == n
    ctx.shard_build(2 * n)                       # as the merging rank: room for one other rank with as many keys
    kb = max(1, int(G).bit_length())             # bits_for(G + 1): the smallest kb with 2^kb >= G + 1
    assert int(keys.max() & np.uint64((1 << kb) - 1)) < G and int(keys.max() >> np.uint64(kb)) < G
    bad = keys.copy()
    bad[n // 2] = np.uint64(((G + 5) << kb) | (G + 9))   # both ids beyond the scene
    with pytest.raises(ValueError, match="outside this scene"):
        ctx.shard_import(0, G, nodes.ctypes.data, n, bad.ctypes.data)
    swapped = keys.copy()
    a, b = int(keys[0] >> np.uint64(kb)), int(keys[0] & np.uint64((1 << kb) - 1))
    swapped[0] = np.uint64((b << kb) | a)               # max << kb | min
    with pytest.raises(ValueError, match="outside this scene"):
        ctx.shard_import(0, G, nodes.ctypes.data, n, swapped.ctypes.data)
    ctx.shard_import(0, G, nodes.ctypes.data, n, keys.ctypes.data)   # the same shard again, untouched: accepted
