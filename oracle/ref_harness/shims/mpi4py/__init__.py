"""Single-process stand-in for mpi4py.MPI.COMM_WORLD (dist_util.py:13)."""


class _World:
    rank = 0
    size = 1

    def Get_rank(self):
        return 0

    def Get_size(self):
        return 1

    def bcast(self, obj, root=0):
        return obj


class MPI:
    COMM_WORLD = _World()
