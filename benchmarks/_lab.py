"""Route a benchmark process through the LAB build of the library (product sources + experiment kernels + the DVD_*
A/B switches; `make -C dvd_amd/csrc lab`).  The product library itself honours no environment variable."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
LAB = os.path.join(ROOT, "benchmarks", "lab", "libdvd_hip_lab.so")


def use_lab():
    from dvd_amd import lib
    lib.use_library(LAB)
    return "lab"


def which(argv=None):
    """`--lab` on the command line selects the lab build; default is the product library."""
    argv = sys.argv if argv is None else argv
    if "--lib" in argv:                      # any other build of the library (benchmarks/lab/alt/: compile-time variants)
        i = argv.index("--lib")
        path = argv[i + 1]
        del argv[i:i + 2]
        from dvd_amd import lib
        lib.use_library(path)
        return os.path.basename(path)
    if "--lab" in argv:
        argv.remove("--lab")
        return use_lab()
    return "product"
