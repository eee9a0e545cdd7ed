"""The C-ABI row-partitioned Ell (include/ellhip_sharded.h) with MORE THAN ONE RANK on the one GPU a test box has: every
rank is a host thread with its own `ellhip_sharded` handle, and the one collective per update goes either through the
library's RCCL call path against an in-process stand-in for librccl (tests/cpp/fake_rccl.cpp, handed over with
ELLHIP_RCCL_PATH -- RCCL itself refuses two ranks per device) or through host-supplied callbacks
(ellhip_sharded_create_custom).  tests/cpp/sharded_ranks_runner.cpp drives direct updates, the two-pass and the
pipelined queue in pieces, a flush between two pipelined runs, a two-pass run right after a pipelined one (the vector is
already exchanged: an all-reduce must not run twice) and a failing cut, and checks every rank against the unsharded
engine (equal blocks: bit for bit, symmetric shards: 1e-12) and the CPU oracle (1e-10).  Symmetric shards take the
pipelined runs in GROUPS (matrix-core products of up to 16 queued cuts over the local trapezoid, one all-reduce of the
group's vectors, the group stage on every rank).  Partition: src/ell.rs:97-137."""
import pytest

pytestmark = pytest.mark.gpu

EQUAL, SYMMETRIC = 0, 1
CASES = [
    # (n, P, partition, depth)
    (192, 2, EQUAL, 1), (192, 3, EQUAL, 8), (2048, 2, EQUAL, 1), (2048, 2, EQUAL, 8), (3072, 3, EQUAL, 8), (4096, 2, EQUAL, 8),
    (512, 2, SYMMETRIC, 8), (2048, 2, SYMMETRIC, 8), (2048, 3, SYMMETRIC, 16), (4096, 2, SYMMETRIC, 16), (4096, 3, SYMMETRIC, 8),
    (2048, 2, SYMMETRIC, 24), (4096, 3, SYMMETRIC, 24),
    # P = 8, the only rank count the driver's node runs: both partitions; symmetric shards at depth 8 (groups end at every 8th
    # cut) and 24 (the pipelined runs in groups of up to 16, one all-reduce per group)
    (4096, 8, EQUAL, 1), (4096, 8, EQUAL, 8), (4096, 8, SYMMETRIC, 8), (4096, 8, SYMMETRIC, 24),
]
MODES = ["rccl", "custom"]


@pytest.fixture(scope="module")
def ranks_results(gpu):
    """Every case in ONE runner process (round 3 started one per case: 26 HIP start-ups, the file took 80 s of the suite)."""
    from cpp_build import build_fake_rccl, build_runner, run_json_lines
    exe = build_runner("sharded_ranks_runner.cpp", "hip+oracle")
    args = []
    for mode in MODES:
        for n, P, partition, depth in CASES:
            args += [mode, str(n), str(P), str(partition), str(depth)]
    return run_json_lines(exe, *args, env={"ELLHIP_RCCL_PATH": build_fake_rccl()}, timeout=900)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("n,P,partition,depth", CASES)
def test_ranks_on_one_gpu(ranks_results, mode, n, P, partition, depth):
    key = f"ranks:{mode}:{n}:{P}:{partition}:{depth}"
    assert key in ranks_results, f"the runner printed no line for {key} (did an earlier case take the process down?)"
    out = ranks_results[key]
    assert out["ok"] is True, out
    assert out["vs_oracle"] <= 1e-10
    if partition == EQUAL:
        assert out["bit_identical"] is True
    else:
        assert out["vs_unsharded"] <= 1e-12
    if mode == "custom":
        # ONE collective per GEMV and never two for one vector: 10 direct + 6 two-pass + (1 prime + 8 commits) + after the
        # flush (0 or 1 re-prime: the prime is dropped only if the flush applied something) + 6 commits + 3 two-pass cuts
        # (cut 30 is already primed and exchanged) + (1 prime + 5 commits; the ones behind the failing cut are issued by
        # the host and are no-ops on the device) = 40 or 41.  The double exchange of cut 30 would make it 41 or 42 AND
        # break the symmetric shards' results (an all-reduce is not idempotent), which the comparisons above catch.
        if partition == EQUAL:
            assert out["collectives"] in (40, 41), out
        else:
            # symmetric shards look ahead in the pipelined runs (ellhip_sharded_queue_run_fused): ONE
            # all-reduce per GROUP of queued cuts (a group ends at an apply pass and at the end of the run), none for a prime:
            # 10 direct + 6 two-pass + 3 two-pass (cut 30 is no longer primed by the run before) + the groups of the three
            # pipelined runs (9, 6 and 6 cuts; 1 or 2 groups each, depending on where the depth's apply passes fall) + ONE
            # all-reduce of a single flag before the first group run (the ranks agree that everybody holds the group buffers)
            assert 23 <= out["collectives"] <= 27, out
