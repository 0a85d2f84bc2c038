"""The parallel collision rule of dapol_build_leaf_nodes (dapol_amd/csrc/kernels_leaf.h: k_leaf_claim_first / k_leaf_settle) as a model
on the CPU: every entity claims its candidate slot with an atomicMin on its INPUT POSITION, an entity whose slot now belongs to an
earlier one draws its next candidate, rounds repeat until nobody moves.  Claim: the fixed point is what the reference's sequential loop
gives (src/dapol/mod.rs:408-441: an entity takes its first candidate that no EARLIER entity holds), including WHICH entity fails when the
candidates run out.  Checked here on adversarially small slot spaces, with the entities of a round visited in random order (the kernel's
threads race; atomicMin makes the outcome of a round order-free, but the test does not rely on that)."""
import random


def sequential(chains, max_tries):
    taken, out = set(), []
    for e, ch in enumerate(chains):
        for k in range(max_tries):
            if ch[k] not in taken:
                taken.add(ch[k])
                out.append(ch[k])
                break
        else:
            return out, e                                  # FailedToMapIndex at entity e; later entities are never looked at
    return out, None


def parallel(chains, max_tries, rng):
    n = len(chains)
    owner = {}                                             # slot -> earliest entity that has claimed it

    def claim(slot, e):                                    # atomicMin; returns the previous owner (None: free)
        prev = owner.get(slot)
        if prev is None or prev > e:
            owner[slot] = e
        return prev

    tries = [1] * n
    for e in rng.sample(range(n), n):
        claim(chains[e][0], e)
    failed = None
    moved = True
    while moved:
        moved = False
        for e in rng.sample(range(n), n):
            if tries[e] > max_tries:                           # parked: out of candidates (it holds no slot; the rounds go on)
                continue
            if owner[chains[e][tries[e] - 1]] == e:
                continue
            moved = True
            while True:
                if tries[e] >= max_tries:
                    failed = e if failed is None else min(failed, e)
                    tries[e] = max_tries + 1
                    break
                tries[e] += 1
                prev = claim(chains[e][tries[e] - 1], e)
                if prev is None or prev > e:
                    break
    if failed is not None:
        return None, failed                                    # the EARLIEST entity that ran out: the one the sequential loop stops at
    return [chains[e][tries[e] - 1] for e in range(n)], None


def test_fixed_point_of_the_claim_settle_rounds_is_the_sequential_result():
    rng = random.Random(2024)
    for case in range(3000):
        n = rng.randint(1, 40)
        slots = rng.randint(max(1, n // 2), 3 * n)          # from hopeless (fewer slots than entities) to roomy
        max_tries = rng.randint(1, 8)
        chains = [[rng.randrange(slots) for _ in range(max_tries)] for _ in range(n)]
        want, want_fail = sequential(chains, max_tries)
        got, got_fail = parallel(chains, max_tries, rng)
        assert got_fail == want_fail, (case, chains)
        if want_fail is None:
            assert got == want, (case, chains)


def test_an_entity_only_ever_walks_a_prefix_of_its_sequential_walk():
    """The invariant behind the claim: a move is forced by an EARLIER entity that holds the slot for good, so no entity ever draws a
    candidate the sequential loop would not have drawn for it."""
    rng = random.Random(7)
    for case in range(500):
        n, slots, max_tries = 30, 40, 10
        chains = [[rng.randrange(slots) for _ in range(max_tries)] for _ in range(n)]
        want, fail = sequential(chains, max_tries)
        if fail is not None:
            continue
        depth = [chains[e].index(want[e]) + 1 if want[e] in chains[e] else max_tries for e in range(n)]
        # re-run the rounds, recording the deepest candidate each entity ever drew
        owner, tries = {}, [1] * n
        for e in rng.sample(range(n), n):
            if owner.get(chains[e][0], n) > e:
                owner[chains[e][0]] = e
        moved = True
        while moved:
            moved = False
            for e in rng.sample(range(n), n):
                while owner[chains[e][tries[e] - 1]] != e:
                    moved = True
                    tries[e] += 1
                    s = chains[e][tries[e] - 1]
                    if owner.get(s, n) > e:
                        owner[s] = e
                assert tries[e] <= depth[e], (case, e)
