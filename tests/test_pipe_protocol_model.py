"""A model of the pipelined AGC / SAM kernel's hand-over protocol (rx_kernels.hip: agc_prep_pipe and the PIPE block of
rx512_kernel): 16 waves of one workgroup, each running the program

    for f in 0 .. F + 1:
        if f < F:   front end + preparation of frame f  -> writes slot f % 3, ready[f % 3] += 1
        g = f - 1:  if 0 <= g < F and this wave takes the duty for g (the first to get here: compare-and-swap on the next
                    unclaimed frame; or, the other policy modelled, g % nvalid == wave): wait ready[g % 3] == nvalid,
                    reset it, (request the first operands,) wait done == g, run the chain of g (reads / writes
                    slot g % 3), done = g + 1
        h = f - 2:  if h >= 0: wait done > h, back end of h (reads slot h % 3)

with arbitrary (random) durations of every phase.  Checked on many random schedules: it terminates (no wave waits for
something that cannot happen), a chain sees every channel's operands and its predecessor's state, a back end sees its
chain's results, no slot is overwritten while a chain or a back end may still read it, and the reset of a ready counter
never swallows an increment that belongs to a later frame."""
import heapq
import random

import pytest


def simulate(nvalid, F, rng, chain_cost, front_cost, back_cost, claim=False):
    ready = [0, 0, 0]
    done = 0
    next_claim = [0]  # claim=True: the chain of frame g goes to the first wave that gets to its duty point (compare-and-swap g -> g + 1)
    t_prep_end = [[None] * F for _ in range(nvalid)]       # [wave][frame]
    t_back = [[(None, None)] * F for _ in range(nvalid)]   # (start, end)
    t_chain = [(None, None)] * F
    ready_owner = [[], [], []]                              # which frames' increments a counter currently holds
    # each wave is a generator of (kind, payload) requests; the scheduler advances the earliest runnable one
    def program(w):
        for f in range(F + 2):
            if f < F:
                yield ("run", front_cost())
                yield ("prep_done", f)
            g = f - 1
            mine = 0 <= g < F and g % nvalid == w
            if claim and 0 <= g < F:
                assert next_claim[0] >= g, "a wave reaches frame g's duty point before g - 1 was taken"
                mine = next_claim[0] == g
                if mine:
                    next_claim[0] = g + 1
            if mine:
                yield ("wait_ready", g)
                yield ("wait_done_eq", g)
                yield ("chain_begin", g)
                yield ("run", chain_cost())
                yield ("chain_end", g)
            h = f - 2
            if h >= 0:
                yield ("wait_done_gt", h)
                yield ("back_begin", h)
                yield ("run", back_cost())
                yield ("back_end", h)

    progs = [program(w) for w in range(nvalid)]
    now = [0.0] * nvalid
    pending = [None] * nvalid            # a blocked request, retried when state changes
    finished = [False] * nvalid
    heap = [(0.0, w) for w in range(nvalid)]
    heapq.heapify(heap)
    blocked = set()
    steps = 0
    while heap:
        steps += 1
        assert steps < 2_000_000
        t, w = heapq.heappop(heap)
        now[w] = max(now[w], t)
        progressed = True
        while progressed:
            req = pending[w]
            if req is None:
                try:
                    req = next(progs[w])
                except StopIteration:
                    finished[w] = True
                    break
            pending[w] = None
            kind, x = req
            if kind == "run":
                now[w] += x
                heapq.heappush(heap, (now[w], w))
                progressed = False
            elif kind == "prep_done":
                f = x
                # the slot f % 3 was last used by frame f - 3: its chain and THIS wave's back end must be over
                if f >= 3:
                    assert t_chain[f - 3][1] is not None and t_chain[f - 3][1] <= now[w], "slot overwritten under a running chain"
                    assert t_back[w][f - 3][1] is not None and t_back[w][f - 3][1] <= now[w], "slot overwritten before its back end read it"
                t_prep_end[w][f] = now[w]
                ready[f % 3] += 1
                ready_owner[f % 3].append(f)
            elif kind == "wait_ready":
                g = x
                if ready[g % 3] >= nvalid:
                    assert ready_owner[g % 3] == [g] * nvalid, "a ready counter mixes frames: %r" % (ready_owner[g % 3],)
                    ready[g % 3] = 0
                    ready_owner[g % 3] = []
                else:
                    pending[w] = req
                    blocked.add(w)
                    progressed = False
            elif kind == "wait_done_eq":
                if done >= x:
                    assert done == x
                else:
                    pending[w] = req
                    blocked.add(w)
                    progressed = False
            elif kind == "wait_done_gt":
                if done > x:
                    pass
                else:
                    pending[w] = req
                    blocked.add(w)
                    progressed = False
            elif kind == "chain_begin":
                g = x
                assert all(t_prep_end[v][g] is not None and t_prep_end[v][g] <= now[w] for v in range(nvalid)), "chain without every channel's operands"
                assert g == 0 or t_chain[g - 1][1] <= now[w], "chain before its predecessor's state"
                t_chain[g] = (now[w], None)
            elif kind == "chain_end":
                g = x
                t_chain[g] = (t_chain[g][0], now[w])
                done = g + 1
            elif kind == "back_begin":
                h = x
                assert t_chain[h][1] is not None and t_chain[h][1] <= now[w], "back end before its chain's results"
                t_back[w][h] = (now[w], None)
            elif kind == "back_end":
                t_back[w][x] = (t_back[w][x][0], now[w])
            # a state change may unblock others: they resume no earlier than now
            if kind in ("prep_done", "chain_end") and blocked:
                for v in list(blocked):
                    blocked.discard(v)
                    heapq.heappush(heap, (max(now[v], now[w]), v))
    assert all(finished), "deadlock: waves %r never finished" % [w for w in range(nvalid) if not finished[w]]
    assert done == F
    return max(now)


@pytest.mark.parametrize("claim", [False, True], ids=["rotating-duty", "claimed-duty"])
@pytest.mark.parametrize("nvalid", [1, 2, 3, 5, 16])
def test_protocol_terminates_and_orders_every_access(nvalid, claim):
    rng = random.Random(1000 + nvalid)
    for trial in range(60):
        F = rng.choice([4, 5, 7, 16, 33])
        style = trial % 4
        if style == 0:    # the measured proportions: chain ~60, front ~47, back ~10 (k cycles)
            costs = (lambda: rng.uniform(55, 65), lambda: rng.uniform(40, 55), lambda: rng.uniform(8, 12))
        elif style == 1:  # a chain much longer than everything else
            costs = (lambda: rng.uniform(200, 400), lambda: rng.uniform(1, 50), lambda: rng.uniform(1, 20))
        elif style == 2:  # a negligible chain, wildly uneven waves
            costs = (lambda: rng.uniform(0.1, 1), lambda: rng.choice([1, 5, 300]), lambda: rng.choice([1, 100]))
        else:             # everything random over three decades
            costs = (lambda: 10 ** rng.uniform(-1, 2), lambda: 10 ** rng.uniform(-1, 2), lambda: 10 ** rng.uniform(-1, 2))
        simulate(nvalid, F, rng, *costs, claim=claim)


def test_two_frames_of_slack_keep_the_duty_waves_lag_off_the_chain_path():
    """the design's argument: with the measured proportions the frame period is set by the waves' own work
    plus their share of the chains, not by chain + front end as with one frame of slack"""
    rng = random.Random(7)
    F = 64
    total = simulate(16, F, rng, lambda: 60.0, lambda: 47.0, lambda: 10.0)
    per_frame = total / F
    assert per_frame < 0.85 * (60.0 + 47.0), per_frame      # clearly below chain + front end
    assert per_frame >= 57.0 + 60.0 / 16 - 1e-9             # and no better than a wave's own work + its share of the duty


# ---- round 4: the synchronous detector behind the AGC -- two stages of the same protocol in a row (rx512_kernel, PSA) ----
def simulate2(nvalid, F, rng, cost_a, cost_chain1, cost_b, cost_chain2, cost_c):
    """Every wave runs, for f in 0 .. F + 3:
         A(f)        front end + AGC preparation  -> slot A[f % 3], readyA[f % 3] += 1
         duty 1      the first wave to get here takes chain1(f - 1): waits readyA == nvalid, resets it, waits doneA == f - 1
         B(f - 2)    waits doneA > f - 2; gain, hand-over -> slot S[(f - 2) % 3], readyS[(f - 2) % 3] += 1
         duty 2      the first wave to get here takes chain2(f - 3): waits readyS == nvalid, resets it, waits doneS == f - 3
         C(f - 4)    waits doneS > f - 4; back end (reads slot S[(f - 4) % 3])"""
    ready = {"A": [0, 0, 0], "S": [0, 0, 0]}
    owner = {"A": [[], [], []], "S": [[], [], []]}
    done = {"A": 0, "S": 0}
    claim = {"A": 0, "S": 0}
    t_a = [[None] * F for _ in range(nvalid)]
    t_b = [[(None, None)] * F for _ in range(nvalid)]
    t_c = [[(None, None)] * F for _ in range(nvalid)]
    t_ch = {"A": [(None, None)] * F, "S": [(None, None)] * F}

    def duty(stage, g):
        mine = False
        if 0 <= g < F:
            assert claim[stage] >= g, "a wave reaches a duty point before the previous frame's was taken"
            mine = claim[stage] == g
            if mine:
                claim[stage] = g + 1
        return mine

    def program(w):
        for f in range(F + 4):
            if f < F:
                yield ("run", cost_a())
                yield ("a_done", f)
            if duty("A", f - 1):
                yield ("wait_ready", ("A", f - 1))
                yield ("wait_done_eq", ("A", f - 1))
                yield ("chain_begin", ("A", f - 1))
                yield ("run", cost_chain1())
                yield ("chain_end", ("A", f - 1))
            if 0 <= f - 2 < F:
                yield ("wait_done_gt", ("A", f - 2))
                yield ("b_begin", f - 2)
                yield ("run", cost_b())
                yield ("b_done", f - 2)
            if duty("S", f - 3):
                yield ("wait_ready", ("S", f - 3))
                yield ("wait_done_eq", ("S", f - 3))
                yield ("chain_begin", ("S", f - 3))
                yield ("run", cost_chain2())
                yield ("chain_end", ("S", f - 3))
            if f - 4 >= 0:
                yield ("wait_done_gt", ("S", f - 4))
                yield ("c_begin", f - 4)
                yield ("run", cost_c())
                yield ("c_end", f - 4)

    progs = [program(w) for w in range(nvalid)]
    now = [0.0] * nvalid
    pending = [None] * nvalid
    finished = [False] * nvalid
    heap = [(0.0, w) for w in range(nvalid)]
    heapq.heapify(heap)
    blocked = set()
    steps = 0
    while heap:
        steps += 1
        assert steps < 4_000_000
        t, w = heapq.heappop(heap)
        now[w] = max(now[w], t)
        progressed = True
        while progressed:
            req = pending[w]
            if req is None:
                try:
                    req = next(progs[w])
                except StopIteration:
                    finished[w] = True
                    break
            pending[w] = None
            kind, x = req
            wake = False
            if kind == "run":
                now[w] += x
                heapq.heappush(heap, (now[w], w))
                progressed = False
            elif kind == "a_done":
                f = x
                if f >= 3:  # slot A[f % 3] last held frame f - 3: its chain and THIS wave's gain stage must be over
                    assert t_ch["A"][f - 3][1] is not None and t_ch["A"][f - 3][1] <= now[w], "AGC slot overwritten under a running chain"
                    assert t_b[w][f - 3][1] is not None and t_b[w][f - 3][1] <= now[w], "AGC slot overwritten before its gain stage read it"
                t_a[w][f] = now[w]
                ready["A"][f % 3] += 1
                owner["A"][f % 3].append(f)
                wake = True
            elif kind == "b_begin":
                fm = x
                assert t_ch["A"][fm][1] is not None and t_ch["A"][fm][1] <= now[w], "gain before its chain's results"
                if fm >= 3:  # slot S[fm % 3] last held frame fm - 3: its PLL and THIS wave's back end must be over
                    assert t_ch["S"][fm - 3][1] is not None and t_ch["S"][fm - 3][1] <= now[w], "PLL slot overwritten under a running loop"
                    assert t_c[w][fm - 3][1] is not None and t_c[w][fm - 3][1] <= now[w], "PLL slot overwritten before its back end read it"
                t_b[w][fm] = (now[w], None)
            elif kind == "b_done":
                fm = x
                t_b[w][fm] = (t_b[w][fm][0], now[w])
                ready["S"][fm % 3] += 1
                owner["S"][fm % 3].append(fm)
                wake = True
            elif kind == "wait_ready":
                st, g = x
                if ready[st][g % 3] >= nvalid:
                    assert owner[st][g % 3] == [g] * nvalid, "a ready counter mixes frames: %r" % (owner[st][g % 3],)
                    ready[st][g % 3] = 0
                    owner[st][g % 3] = []
                else:
                    pending[w] = req
                    blocked.add(w)
                    progressed = False
            elif kind == "wait_done_eq":
                st, g = x
                if done[st] >= g:
                    assert done[st] == g
                else:
                    pending[w] = req
                    blocked.add(w)
                    progressed = False
            elif kind == "wait_done_gt":
                st, g = x
                if not done[st] > g:
                    pending[w] = req
                    blocked.add(w)
                    progressed = False
            elif kind == "chain_begin":
                st, g = x
                src = t_a if st == "A" else None
                if st == "A":
                    assert all(t_a[v][g] is not None and t_a[v][g] <= now[w] for v in range(nvalid)), "AGC chain without every channel's operands"
                else:
                    assert all(t_b[v][g][1] is not None and t_b[v][g][1] <= now[w] for v in range(nvalid)), "PLL without every channel's samples"
                assert g == 0 or t_ch[st][g - 1][1] <= now[w], "chain before its predecessor's state"
                t_ch[st][g] = (now[w], None)
            elif kind == "chain_end":
                st, g = x
                t_ch[st][g] = (t_ch[st][g][0], now[w])
                done[st] = g + 1
                wake = True
            elif kind == "c_begin":
                fb = x
                assert t_ch["S"][fb][1] is not None and t_ch["S"][fb][1] <= now[w], "back end before its PLL's audio"
                t_c[w][fb] = (now[w], None)
            elif kind == "c_end":
                t_c[w][x] = (t_c[w][x][0], now[w])
            if wake and blocked:
                for v in list(blocked):
                    blocked.discard(v)
                    heapq.heappush(heap, (max(now[v], now[w]), v))
    assert all(finished), "deadlock: waves %r never finished" % [w for w in range(nvalid) if not finished[w]]
    assert done["A"] == F and done["S"] == F
    return max(now)


@pytest.mark.parametrize("nvalid", [1, 2, 3, 5, 16])
def test_two_stage_protocol_terminates_and_orders_every_access(nvalid):
    rng = random.Random(2000 + nvalid)
    for trial in range(50):
        F = rng.choice([4, 5, 6, 9, 16, 33])
        style = trial % 4
        if style == 0:    # the measured proportions (k cycles): front + preparation 40, AGC chain 65, gain 5, PLL 150, back end 12
            costs = (lambda: rng.uniform(35, 45), lambda: rng.uniform(60, 70), lambda: rng.uniform(4, 6), lambda: rng.uniform(140, 160), lambda: rng.uniform(10, 14))
        elif style == 1:  # the first chain the long one
            costs = (lambda: rng.uniform(1, 50), lambda: rng.uniform(200, 400), lambda: rng.uniform(1, 5), lambda: rng.uniform(1, 30), lambda: rng.uniform(1, 20))
        elif style == 2:  # negligible chains, wildly uneven waves
            costs = (lambda: rng.choice([1, 5, 300]), lambda: rng.uniform(0.1, 1), lambda: rng.choice([1, 50]), lambda: rng.uniform(0.1, 1), lambda: rng.choice([1, 100]))
        else:             # everything random over three decades
            costs = tuple((lambda: 10 ** rng.uniform(-1, 2)) for _ in range(5))
        simulate2(nvalid, F, rng, *costs)


def test_two_stage_period_is_the_longer_chain():
    """with the measured proportions the frame period is the PLL's (the longer chain), not the sum of the two chains:
    the point of giving each chain a duty wave of its own"""
    rng = random.Random(9)
    F = 48
    total = simulate2(16, F, rng, lambda: 40.0, lambda: 65.0, lambda: 5.0, lambda: 150.0, lambda: 12.0)
    per_frame = total / F
    assert per_frame < 0.80 * (65.0 + 150.0), per_frame
    assert per_frame >= 150.0 - 1e-9
