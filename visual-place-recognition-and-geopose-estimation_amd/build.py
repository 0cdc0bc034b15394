"""Build libvpr_amd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def library_path() -> str:
    return os.path.join(_HERE, "libvpr_amd.so")


def build_library(force: bool = False, jobs: int = 4) -> str:
    """Run the csrc Makefile; returns the path of the shared library."""
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.run(["make", "-C", csrc, "clean"], check=True, stdout=subprocess.DEVNULL)
    proc = subprocess.run(["make", "-C", csrc, f"-j{jobs}"], stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc build of libvpr_amd.so failed:\n" + proc.stdout)
    if not os.path.exists(library_path()):
        raise RuntimeError("build finished but libvpr_amd.so is missing")
    return library_path()
