"""Rank program of tests/test_dist2d.py::test_the_memory_verdict_is_collective (CPU, gloo).  Only the rank named in
MADQP_TEST_SMALL_RANK sees a small "free device memory" figure (MADQP_TEST_MEM_FREE, read by tests/csrc/dist_cpu.cpp's
dop_mem_free): the create of csrc/dist_core.inc must refuse on EVERY rank (a rank that went on alone would
wait for the others in the communicator set-up or its first collective for ever)."""
import ctypes as C
import json
import os
import sys

import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from madqp_jl_amd.dist2d import HostStagedComm  # noqa: E402


def main():
    out, P, Q, n, nb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    if rank == int(os.environ["MADQP_TEST_SMALL_RANK"]):
        os.environ["MADQP_TEST_MEM_FREE"] = os.environ["MADQP_TEST_SMALL_BYTES"]
    lib = C.CDLL(os.path.join(ROOT, "tests", "_build", "libmadqp_dist_cpuref.so"))
    lib.madqp_distcpu_last_error.restype = C.c_char_p
    comm = HostStagedComm(P, Q)
    h = C.c_void_p()
    rc = lib.madqp_distcpu_create(rank, world, P, Q, C.c_int64(n), C.c_int64(nb), C.byref(comm.ops), C.byref(h))
    rec = dict(rank=rank, rc=rc, handle=bool(h.value), err=lib.madqp_distcpu_last_error().decode(), comm_error=comm.error)
    # a second create with room everywhere must work on the same process group: nobody is stuck in a collective
    os.environ.pop("MADQP_TEST_MEM_FREE", None)
    h2 = C.c_void_p()
    rec["rc_again"] = lib.madqp_distcpu_create(rank, world, P, Q, C.c_int64(n), C.c_int64(nb), C.byref(comm.ops), C.byref(h2))
    if h2.value:
        lib.madqp_distcpu_destroy(h2)
    json.dump(rec, open(f"{out}.{rank}", "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
