"""Rank program of test_slab.py::test_rccl_id_travels_through_the_rendezvous_store: the hand-over
NativeComm.over_store() uses, with a stand-in for ncclGetUniqueId (no GPU here)."""
import os
import sys

from yalla_amd.slab import COMM_ID_BYTES, NativeComm

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
made = bytes((7 * k + 3) % 256 for k in range(COMM_ID_BYTES))
ident, store = NativeComm._id_over_store(rank, world, "yalla_test_id", lambda: made)
assert ident == made, "the id arrived changed"
store.add("yalla_test_seen", 1)
if rank == 0:   # (rank 0 may be the store's server: stay until everyone has read)
    import time
    for _ in range(600):
        if int(store.add("yalla_test_seen", 0)) >= world:
            break
        time.sleep(0.1)
with open(sys.argv[1] + f".{rank}", "w") as out:
    out.write("ok")
