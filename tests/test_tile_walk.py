"""Host-side model of the persistent GEMM's tile walk (csrc/gemm_pers.hip: make_sched, loader_role, launch_gemm_pers): the workgroups of one (logical)
XCD take their first tile statically and every further one from the XCD's ticket counter; the holder of the launch's LAST ticket re-arms the counter
(an agent-scope atomic exchange since round 4).  Random interleavings of the workgroups' draws, several launches of different shapes back to back, a
capped grid: every tile is walked exactly once, every workgroup stops, and the counters are back at zero when a launch ends.  The static deal (the A/B
form the stress runs used to clear the tickets) walks the same set.  CPU only — the arithmetic of the protocol, not the kernel."""
import random


def sched(total, G, xcd):
    """make_sched: the XCD's contiguous chunk of logical tile ids and its workgroup count"""
    q8, r8 = total >> 3, total & 7
    cnt = q8 + (1 if xcd < r8 else 0)
    base = xcd * (q8 + 1) if xcd < r8 else r8 * (q8 + 1) + (xcd - r8) * q8
    nwg = (G >> 3) + (1 if xcd < (G & 7) else 0)
    return cnt, base, nwg


def run_launch(total, G, counters, rng, dynamic=True):
    walked = []
    for xcd in range(8):
        cnt, base, nwg = sched(total, G, xcd)
        assert nwg <= cnt                                   # the launcher keeps G <= total
        pending = list(range(nwg))                          # local tile index each workgroup is about to process (its slot first)
        rearmed = False
        while pending:
            i = rng.randrange(len(pending))                 # any interleaving of the workgroups' K-tile-0 hand-overs
            li = pending[i]
            walked.append(base + li)
            if dynamic:
                ticket = counters[xcd]                      # atomicAdd's old value
                counters[xcd] += 1
                if ticket == cnt - 1:                       # the launch's last ticket on this XCD: atomicExch(ctr, 0)
                    assert not rearmed
                    counters[xcd] = 0
                    rearmed = True
                nxt = nwg + ticket
            else:
                nxt = li + nwg                              # static deal
            if nxt < cnt:
                pending[i] = nxt
            else:
                pending.pop(i)
        if dynamic:
            assert rearmed and counters[xcd] == 0, (xcd, total, G, counters[xcd])
    return walked


def test_every_tile_exactly_once_and_counters_rearmed():
    rng = random.Random(7)
    counters = [0] * 8
    shapes = [(6128, 256), (4608, 256), (1536, 256), (257, 256), (256, 256), (9, 9), (8, 8), (300, 104), (6128, 9), (1533, 256), (24, 24), (6144, 192)]
    for total, G in shapes * 3:
        assert sorted(run_launch(total, G, counters, rng)) == list(range(total)), (total, G)
        assert counters == [0] * 8


def test_static_deal_walks_the_same_tiles():
    rng = random.Random(11)
    for total, G in [(6128, 256), (257, 256), (300, 104), (24, 24)]:
        assert sorted(run_launch(total, G, [0] * 8, rng, dynamic=False)) == list(range(total))
