"""-m gpu: `import limap_amd` + a triangulation BEFORE `import torch` must leave torch.cuda usable (one HIP runtime per
process: torch's bundled libamdhip64 is preloaded before anything that links the extension -- _capi._preload_torch_hip_runtime)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import sys
sys.path.insert(0, %r)
from limap_amd import synthetic as syn, triangulation as tri
sc = syn.make_scene(n_views=10, n_segs=60, n_neighbors=4, seed=1)
T = tri.GlobalLineTriangulator(syn.default_triangulation_cfg())
T.SetRanges(sc.ranges)
T.InitArrays(sc.img_ids, sc.kvec, sc.qvec, sc.tvec, [sc.segs_of(j) for j in range(sc.n_images)])
for i in sc.img_ids:
    T.TriangulateImage(int(i), sc.matches_of(int(i)))
n = len(T.ComputeLineTracks())
assert "torch" not in sys.modules
import torch
x = torch.ones(8, device="cuda")
print("OK", n, float(x.sum().item()))
'''


def test_torch_after_limap_amd(gpu_lib):
    p = subprocess.run([sys.executable, "-c", CODE % ROOT], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "OK" in p.stdout, p.stdout[-500:] + p.stderr[-1500:]
    assert p.stdout.strip().splitlines()[-1].split()[-1] == "8.0"
