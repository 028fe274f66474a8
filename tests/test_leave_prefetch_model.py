"""CPU restatement of the ring arithmetic behind r05's leave-prefetch in the throughput tracking kernel (csrc/kernels_track_block.hpp,
GYP_EXPERIMENT_LEAVE_PF): wavefront 3 fetches, during millisecond ms, the ring entries that costas_update's fetch_leaving will want in
millisecond ms + 1, from positions it derives from the STEP COUNT alone (n1 = n_first + (ms - ms_first) + 1).  That is only right if the
incrementally updated positions of the loop state (pos_e, pos_p) always equal n % 250 and n % 1000 -- the prologue derives them that way at
every launch -- and if the entry read a millisecond early is not written in between.  Both are checked here for every launch cut."""
from hypothesis import given, settings, strategies as st

K_LOCK, K_PEAKS = 250, 1000      # kLockWindow, kPeakHistory (csrc/kernels_common.hpp)


def fetch_leaving_positions(n, pos_e, pos_p):
    """fetch_leaving at the start of the update of the millisecond with n steps before it: (error slot, peak slot) or None."""
    if n < K_LOCK:
        return None
    pos_leave = pos_p - K_LOCK if pos_p >= K_LOCK else pos_p - K_LOCK + K_PEAKS
    return pos_e, pos_leave


def prefetch_positions(n1):
    """What wavefront 3 computes a millisecond ahead from the step count n1 of the NEXT millisecond."""
    if n1 < K_LOCK:
        return None
    pe, pp = n1 % K_LOCK, n1 % K_PEAKS
    return pe, (pp - K_LOCK if pp >= K_LOCK else pp - K_LOCK + K_PEAKS)


@settings(max_examples=50, deadline=None)
@given(st.integers(0, 5000), st.lists(st.integers(1, 700), min_size=1, max_size=6))
def test_prefetched_slots_are_the_ones_the_next_update_reads_and_nobody_writes_them_meanwhile(n_start, launches):
    n = n_start
    for length in launches:                       # a block cut into launches: the prologue re-derives the positions from n_steps
        pos_e, pos_p = n % K_LOCK, n % K_PEAKS
        n_first = n
        for ms in range(length):
            assert pos_e == n % K_LOCK and pos_p == n % K_PEAKS
            want = fetch_leaving_positions(n, pos_e, pos_p)
            if ms > 0:
                assert prefetched == want          # what was fetched during ms - 1 is what this update needs
            # this millisecond's update writes the peak at pos_p and the error at pos_e ...
            written = (pos_e, pos_p)
            # ... while wavefront 3 reads the slots of ms + 1
            prefetched = prefetch_positions(n_first + ms + 1)
            if prefetched is not None:
                assert prefetched[0] != written[0]                    # the error slot read is not the one being written
                assert prefetched[1] != written[1]                    # nor the peak slot
            pos_e = 0 if pos_e + 1 == K_LOCK else pos_e + 1
            pos_p = 0 if pos_p + 1 == K_PEAKS else pos_p + 1
            n += 1
