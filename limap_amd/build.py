"""In-tree build of the HIP extension (liblimap_amd.so) for gfx950.  hipcc cross-compiles without a GPU."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "liblimap_amd.so")


def build_extension(force=False, verbose=False):
    """`make -C limap_amd/csrc` (hipcc --offload-arch=gfx950 for lt_kernels.hip, g++ for the host side)."""
    cmd = ["make", "-C", CSRC]
    if force:
        cmd.append("-B")
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building liblimap_amd.so failed")
    return LIB_PATH
