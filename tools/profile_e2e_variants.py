"""Developer tool: the end-to-end call sequence under the conditions bench.py measures it in (torch imported and initialised,
the Python track list built, a post-processing chain between repetitions), one factor at a time.
   python tools/profile_e2e_variants.py [torch] [torch_import] [gomp] [sysgomp] [tracks] [post]"""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
flags = set(sys.argv[1:])
if "gomp" in flags:  # torch's bundled libgomp (an older build with the system one's soname) without torch: whoever loads first wins
    import ctypes, importlib.util
    _t = os.path.dirname(importlib.util.find_spec("torch").origin)
    ctypes.CDLL(os.path.join(_t, "lib", "libgomp.so"), mode=ctypes.RTLD_GLOBAL)
for _f, _libs in (("roctracer", ("libroctracer64.so", "librocprofiler-register.so")), ("rocblas", ("librocblas.so", "libMIOpen.so", "libhipblaslt.so")),
                  ("c10", ("libc10.so",)), ("torch_cpu", ("libtorch_cpu.so",)), ("torch_hip", ("libtorch_hip.so",))):
    if _f in flags:  # parts of what `import torch` loads, without torch: which one slows the HIP calls of this library down?
        import ctypes, importlib.util
        _t = os.path.dirname(importlib.util.find_spec("torch").origin)
        ctypes.CDLL(os.path.join(_t, "lib", "libamdhip64.so"), mode=ctypes.RTLD_GLOBAL)
        for _l in _libs:
            ctypes.CDLL(os.path.join(_t, "lib", _l), mode=ctypes.RTLD_GLOBAL)
if "sysgomp" in flags:  # the system's libgomp first, then torch: torch runs on the system's
    import ctypes
    ctypes.CDLL("libgomp.so.1", mode=ctypes.RTLD_GLOBAL)
if "torch" in flags or "torch_import" in flags or "torch_init" in flags:
    import torch
    if "torch_import" not in flags:
        torch.cuda.init()
    if "torch" in flags:
        x = torch.zeros(1 << 20, device="cuda:0")
        torch.cuda.synchronize()
if "pin" in flags or "pin_other" in flags:  # confine the process to one NUMA node (the one the main thread is on, or the other one)
    cpu = int(open("/proc/self/stat").read().split()[38])
    nodes = {}
    for d in os.listdir("/sys/devices/system/node"):
        if d.startswith("node") and d[4:].isdigit():
            cl = open(f"/sys/devices/system/node/{d}/cpulist").read().strip()
            cpus = set()
            for part in cl.split(","):
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
            nodes[int(d[4:])] = cpus
    mine = [n for n, c in nodes.items() if cpu in c][0]
    pick = mine if "pin" in flags else [n for n in nodes if n != mine][0]
    os.sched_setaffinity(0, nodes[pick])
import numpy as np
from limap_amd import synthetic as syn, triangulation as tri, merging
sc = syn.make_scene(n_views=100, n_segs=500, n_neighbors=20, seed=0)
cfg = syn.default_triangulation_cfg()
matches = {int(i): sc.matches_of(int(i)) for i in sc.img_ids}
segs_list = [sc.segs_of(j) for j in range(sc.n_images)]
res = []
if "freeze" in flags:  # what was imported so far leaves the collector's generations: a collection no longer walks torch's objects
    gc.collect(); gc.freeze()
for rep in range(7):
    if "nogc" not in flags:
        gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    T = tri.GlobalLineTriangulator(cfg)
    T.SetRanges(sc.ranges)
    T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, segs_list)
    t1 = time.perf_counter()
    for i in sc.img_ids:
        T.TriangulateImage(int(i), matches[int(i)])
    t2 = time.perf_counter()
    if "tracks" in flags:
        tr = T.ComputeLineTracks()
    else:
        T.context().compute_tracks()
    t3 = time.perf_counter()
    if "post" in flags:
        ts = merging.TrackSet.from_triangulator(T)
        ts.filter_by_reprojection(8.0, 5.0).remerge(dict(linker2d={}, linker3d={})) if False else ts.filter_by_reprojection(8.0, 5.0)
        ts.filter_by_sensitivity(75.0, 3).filter_by_overlap(0.5, 3)
        del ts
    gc.enable()
    tm = T.timers()
    res.append((t1 - t0, t2 - t1, t3 - t2, t3 - t0))
    del T
r = 1e3 * np.median(np.array(res[2:]), axis=0)
print(sorted(flags), "ctor_init %.2f buffer %.2f compute_tracks %.2f total %.2f" % tuple(r), "[upload %.2f run %.2f tail %.2f]" % (tm["upload"], tm["run"], tm["tail"]))
