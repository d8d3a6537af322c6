"""Write a synthetic workload as the reference's input files: lower-triangular, column-sorted
MatrixMarket (README.md:31) + an ordering file (dimension, then n entries).
    python tools/make_mtx.py small3d /tmp/small3d
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from parsy_bench_amd import matrices as M  # noqa: E402


def write(name, prefix):
    A, perm = M.workload(name)
    with open(prefix + ".mtx", "w") as f:
        f.write("%%MatrixMarket matrix coordinate real symmetric\n")
        f.write(f"{A.n} {A.n} {A.nnz}\n")
        for j in range(A.n):
            for k in range(A.Ap[j], A.Ap[j + 1]):
                f.write(f"{A.Ai[k] + 1} {j + 1} {A.Ax[k]:.17g}\n")
    with open(prefix + ".ord", "w") as f:
        f.write(f"{A.n}\n")
        f.write("\n".join(str(int(p)) for p in perm) + "\n")
    return prefix + ".mtx", prefix + ".ord"


if __name__ == "__main__":
    print(*write(sys.argv[1], sys.argv[2]))
