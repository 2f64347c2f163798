"""The point-to-point schedules of csrc/comm.hip replayed for EVERY rank of R = 1, 2, 3, 8 on the CPU (ocn_comm_schedule is a pure host
function; the library executes exactly these lists inside one RCCL group).  RCCL pairs the k-th send of rank a to rank b with the
k-th receive of b from a: every send must meet a receive, in the buffer the choreography means.  This is what a one-rank GPU box
cannot validate (the R = 2 receive-order swap, the peer choice of the plane exchange, the all-to-all peer loop).
Reference semantics: halo_communication.jl:100-150 (send / recv tags per side), distributed_architectures.jl:386-429 (neighbours wrap),
distributed_transpose.jl:185-191 (Alltoallv! with equal counts)."""
import ctypes as C

import pytest


@pytest.fixture(scope="module")
def L():
    import oceananigans_jl_amd as ocn
    ocn._lib.lib()
    return ocn._lib


def schedule(L, kind, rank, R, self_via_rccl=False):
    ops = (L.CCommOp * 256)()
    n = C.c_int32()
    L.call("ocn_comm_schedule", kind, rank, R, int(self_via_rccl), ops, 256, C.byref(n))
    return [(ops[q].is_recv, ops[q].peer, ops[q].slot) for q in range(n.value)]


def deliver(L, kind, R, self_via_rccl=False):
    """{(receiver, recv slot): (sender, send slot)} after pairing per (sender, receiver) in issue order; asserts nothing is left over"""
    sends, recvs = {}, {}
    for r in range(R):
        for is_recv, peer, slot in schedule(L, kind, r, R, self_via_rccl):
            assert 0 <= peer < R
            if is_recv:
                recvs.setdefault((peer, r), []).append(slot)
            else:
                sends.setdefault((r, peer), []).append(slot)
    assert sends.keys() == recvs.keys(), f"unmatched peers: {sends.keys() ^ recvs.keys()}"
    out = {}
    for pair, ss in sends.items():
        assert len(ss) == len(recvs[pair]), f"{pair}: {len(ss)} sends, {len(recvs[pair])} receives"
        for s_slot, r_slot in zip(ss, recvs[pair]):
            assert (pair[1], r_slot) not in out
            out[(pair[1], r_slot)] = (pair[0], s_slot)
    return out


@pytest.mark.parametrize("R", [2, 3, 8])
def test_strip_exchange_pairs_west_with_east(L, R):
    """my send_west (slot 0) must land in my WEST neighbour's recv_east (slot 3), my send_east (slot 1) in my EAST neighbour's recv_west
    (slot 2) -- also when both neighbours are the same rank (R = 2)"""
    got = deliver(L, L.SCHED_STRIPS, R)
    assert len(got) == 2 * R
    for r in range(R):
        assert got[((r - 1) % R, 3)] == (r, 0)
        assert got[((r + 1) % R, 2)] == (r, 1)


def test_one_rank_strip_exchange(L):
    assert schedule(L, L.SCHED_STRIPS, 0, 1) == []                       # device copies, no RCCL
    got = deliver(L, L.SCHED_STRIPS, 1, self_via_rccl=True)               # a rank talking to itself through RCCL
    assert got == {(0, 3): (0, 0), (0, 2): (0, 1)}


@pytest.mark.parametrize("R", [2, 3, 8])
def test_plane_exchange_peers(L, R):
    """east: I receive my EAST neighbour's west interior plane, so I send mine to my WEST neighbour; west: mirror image"""
    for kind, to, frm in ((L.SCHED_PLANE_EAST, -1, +1), (L.SCHED_PLANE_WEST, +1, -1)):
        got = deliver(L, kind, R)
        assert len(got) == R
        for r in range(R):
            assert got[(r, 1)] == ((r + frm) % R, 0)
            assert schedule(L, kind, r, R)[0] == (0, (r + to) % R, 0)


@pytest.mark.parametrize("R", [1, 2, 3, 8])
@pytest.mark.parametrize("self_via_rccl", [False, True])
def test_all_to_all_delivers_chunk_d_to_rank_d(L, R, self_via_rccl):
    """chunk d of rank s's send buffer arrives as chunk s of rank d's receive buffer: the schedule names chunks by PEER on both sides, so
    the pairing (receiver d, slot s) <- (sender s, slot d) is what Alltoallv! with equal counts does"""
    got = deliver(L, L.SCHED_ALL_TO_ALL, R, self_via_rccl)
    expect = {(d, s): (s, d) for s in range(R) for d in range(R) if self_via_rccl or s != d}
    assert got == expect


@pytest.mark.parametrize("R", [1, 2, 3, 8])
@pytest.mark.parametrize("self_via_rccl", [False, True])
def test_all_gather_delivers_every_chunk_to_every_rank(L, R, self_via_rccl):
    """MPI.Allgather with equal counts as R - 1 direct transfers: rank s's one send buffer (slot 0) arrives as chunk s (slot s) of every
    other rank (the xGMI links of a node are point-to-point: one transfer per link instead of the collective's ring)"""
    got = deliver(L, L.SCHED_ALL_GATHER, R, self_via_rccl)
    expect = {(d, s): (s, 0) for s in range(R) for d in range(R) if self_via_rccl or s != d}
    assert got == expect


def test_schedule_argument_validation(L):
    import oceananigans_jl_amd as ocn
    ops = (L.CCommOp * 4)()
    n = C.c_int32()
    with pytest.raises(ocn.OcnError, match="unknown kind"):
        L.call("ocn_comm_schedule", 9, 0, 2, 0, ops, 4, C.byref(n))
    with pytest.raises(ocn.OcnError, match="rank 5 of 2"):
        L.call("ocn_comm_schedule", 0, 5, 2, 0, ops, 4, C.byref(n))
    with pytest.raises(ocn.OcnError, match="do not fit"):
        L.call("ocn_comm_schedule", L.SCHED_ALL_TO_ALL, 0, 8, 0, ops, 4, C.byref(n))
    assert n.value == 14
