"""The two-queue halo choreography of a partitioned rank (tetsim_halo.hip: enqueue_phase_a / enqueue_phase_b / flush_v), checked as
a MODEL on the CPU: every interleaving of the queues of two neighbouring ranks is explored, and in every one of them

  * nobody deadlocks (each `wait` finds its word raised eventually, each transfer meets its partner),
  * no word is raised twice before it is consumed (the words are binary semaphores),
  * every kernel reads exactly the version of every buffer it is meant to read, for its whole duration, and nobody writes a
    buffer somebody else is still reading or writing.

The model restates the order of submission by hand -- it is not generated from the C++ -- so it pins the DESIGN:
what runs on which queue, which kernel raises which word as it starts, which wait sits where, and what the end of a call adds.
Queues are in order (an operation starts when its predecessor on the queue has ended); a kernel is two events, START and END:
reads last from START to END, writes too (conservatively), a raise happens at START.  The transfer of substep s is a rendezvous
of the two ranks' halo queues (grouped send + receive: RCCL completes both directions together)."""
import pytest

# buffers of one rank; versions count substeps: entering substep s every prediction has version s, the partial sums of substep s
# have version s + 1, the particle passes of substep s take the predictions to version s + 1
PRED_B, PRED_I, PRED_G, PART_H, PART_I = "pred boundary", "pred interior", "pred ghost", "partial halo-side", "partial interior"


def rank_program(calls):
    """The operations one rank submits for `calls` = [n0, n1, ...] (substeps per tetsim_step_n call), per queue, in submission order.
    A NEGATIVE entry -n is a call of n substeps with a new dt: ensure_prediction puts the re-prediction and one more exchange in front.
    op = (name, reads {buf: version}, writes {buf: new version}, raises word or None, waits word or None, transfer id or None,
          join = number of halo-queue operations that must have ended before this one starts, or None)"""
    main, halo = [], []
    s = 0
    for call, n in enumerate(calls):
        if n < 0:
            n = -n
            # ensure_prediction: the main queue goes behind everything the halo queue has been given (an event), re-predicts ALL its
            # particles, and hands over through a word of its own; the halo queue then repeats the exchange with the new predictions
            main.append(("repredict", {PRED_B: s, PRED_I: s}, {PRED_B: s, PRED_I: s}, None, None, None, len(halo)))
            main.append(("signal D", {}, {}, "D", None, None, None))
            halo.append(("wait D", {}, {}, None, "D", None, None))
            halo.append(("X(refresh %d)" % call, {PRED_B: s}, {}, None, None, ("refresh", call, s), None))
        v_pending = False
        for _ in range(n):
            # enqueue_phase_a: interior tiles first (they raise the previous substep's V), then the halo queue's wait for it
            main.append(("T_int(%d)" % s, {PRED_I: s}, {PART_I: s + 1}, "V" if v_pending else None, None, None, None))
            if v_pending:
                halo.append(("wait V(%d)" % (s - 1), {}, {}, None, "V", None, None))
            halo.append(("T_H(%d)" % s, {PRED_B: s, PRED_I: s, PRED_G: s}, {PART_H: s + 1}, None, None, None, None))
            halo.append(("P_b(%d)" % s, {PART_H: s + 1}, {PRED_B: s + 1}, "G", None, None, None))
            main.append(("wait G(%d)" % s, {}, {}, None, "G", None, None))
            main.append(("P_i(%d)" % s, {PART_H: s + 1, PART_I: s + 1}, {PRED_I: s + 1}, None, None, None, None))
            v_pending = True
            # enqueue_phase_b: the transfer (reads our boundary predictions, writes the PEER's ghosts)
            halo.append(("X(%d)" % s, {PRED_B: s + 1}, {}, None, None, ("substep", s, s + 1), None))
            s += 1
        # flush_v: the last substep's V as operations of their own
        main.append(("signal V(%d)" % (s - 1), {}, {}, "V", None, None, None))
        halo.append(("wait V(%d)" % (s - 1), {}, {}, None, "V", None, None))
    return main, halo


class Violation(Exception):
    pass


def explore(calls, mutate=None):
    """Depth-first search over all interleavings of 2 ranks x 2 queues.  Returns the number of distinct states visited."""
    progs = [rank_program(calls), rank_program(calls)]
    if mutate:
        progs = [mutate(p) for p in progs]
    queues = [progs[0][0], progs[0][1], progs[1][0], progs[1][1]]   # rank = q // 2
    total = sum(abs(n) for n in calls)
    bufs = (PRED_B, PRED_I, PRED_G, PART_H, PART_I)
    version0 = tuple((0,) * len(bufs) for _ in range(2))           # [rank][buffer]
    # state: (pc per queue, running flag per queue, words per rank (G, V), versions)
    start = ((0, 0, 0, 0), (False,) * 4, ((0, 0, 0), (0, 0, 0)), version0)
    seen, stack = {start}, [start]
    bi = {b: i for i, b in enumerate(bufs)}
    wi = {"G": 0, "V": 1, "D": 2}

    def active(state):   # (queue, op) of every kernel between START and END
        pcs, running = state[0], state[1]
        return [(q, queues[q][pcs[q]]) for q in range(4) if running[q]]

    while stack:
        state = stack.pop()
        pcs, running, words, versions = state
        if all(pcs[q] == len(queues[q]) for q in range(4)):
            if any(w for r in words for w in r):
                raise Violation("a word is still raised at the end: %r" % (words,))
            if any(versions[r][bi[b]] != total for r in range(2) for b in (PRED_B, PRED_I, PRED_G)):
                raise Violation("final versions %r" % (versions,))
            continue
        moves = []
        for q in range(4):
            if pcs[q] == len(queues[q]):
                continue
            name, reads, writes, raises, waits, xfer, join = queues[q][pcs[q]]
            r = q // 2
            if not running[q]:
                # ---- START
                if waits is not None and not words[r][wi[waits]]:
                    continue                                   # the wait kernel spins: model it as "cannot complete yet"
                if join is not None and pcs[r * 2 + 1] < join:
                    continue                                   # behind an event recorded on the halo queue
                if xfer is not None:
                    peer_q = (1 - r) * 2 + 1                   # rendezvous: the peer's halo queue must be at the same transfer
                    if pcs[peer_q] == len(queues[peer_q]) or queues[peer_q][pcs[peer_q]][5] != xfer or running[peer_q]:
                        continue
                    if r == 1:
                        continue                               # (the pair moves as ONE step, taken from rank 0's side)
                moves.append(("start", q))
            else:
                moves.append(("end", q))
        if not moves:
            raise Violation("deadlock at %s" % [queues[q][pcs[q]][0] if pcs[q] < len(queues[q]) else "-" for q in range(4)])
        for kind, q in moves:
            name, reads, writes, raises, waits, xfer, join = queues[q][pcs[q]]
            r = q // 2
            npcs, nrun, nwords, nver = list(pcs), list(running), [list(w) for w in words], [list(v) for v in versions]
            if kind == "start":
                if xfer is not None:                           # both directions at once; atomic (START and END together)
                    for rr in (0, 1):
                        if versions[rr][bi[PRED_B]] != xfer[2]:
                            raise Violation("%s sends boundary predictions of version %d" % (name, versions[rr][bi[PRED_B]]))
                    for rr in (0, 1):                          # the receiver's ghosts and the sender's boundary particles must not be in use
                        for aq, op in active(state):
                            if aq // 2 == rr and (PRED_G in op[1] or PRED_B in op[2]):
                                raise Violation("%s while %s is running on rank %d" % (name, op[0], rr))
                        nver[rr][bi[PRED_G]] = xfer[2]
                    npcs[q] += 1
                    npcs[(1 - r) * 2 + 1] += 1
                else:
                    if waits is not None:
                        nwords[r][wi[waits]] = 0               # consumed (wait kernels are modelled as atomic)
                        npcs[q] += 1
                    else:
                        if raises is not None:
                            if words[r][wi[raises]]:
                                raise Violation("%s raises %s twice" % (name, raises))
                            nwords[r][wi[raises]] = 1
                        for b, want in reads.items():
                            if versions[r][bi[b]] != want:
                                raise Violation("%s reads %s at version %d, wants %d" % (name, b, versions[r][bi[b]], want))
                        for aq, op in active(state):           # nobody (of this rank) may be writing what we read or write, or reading what we write
                            if aq // 2 != r:
                                continue
                            for b in op[2]:
                                if b in reads or b in writes:
                                    raise Violation("%s starts while %s writes %s" % (name, op[0], b))
                            for b in op[1]:
                                if b in writes:
                                    raise Violation("%s would write %s while %s reads it" % (name, b, op[0]))
                        if not reads and not writes:           # a signal kernel: atomic
                            npcs[q] += 1
                        else:
                            nrun[q] = True
            else:
                for b, new in writes.items():
                    nver[r][bi[b]] = new
                nrun[q] = False
                npcs[q] += 1
            nxt = (tuple(npcs), tuple(nrun), tuple(tuple(w) for w in nwords), tuple(tuple(v) for v in nver))
            if nxt not in seen:
                seen.add(nxt)
                stack.append(nxt)
    return len(seen)


@pytest.mark.parametrize("calls", [[1], [2], [3], [1, 1], [2, 1, 2], [4], [2, -2], [1, -1, -3, 2]])
def test_every_interleaving_is_live_and_race_free(calls):
    assert explore(calls) > 10 * sum(abs(n) for n in calls)


def _drop(pred):
    def mutate(prog):
        return tuple([op for op in queue if not pred(op)] for queue in prog)
    return mutate


def _edit(fn):
    def mutate(prog):
        return tuple([fn(op) for op in queue] for queue in prog)
    return mutate


@pytest.mark.parametrize("what,mutate", [
    # the halo-side tiles of the next substep read interior particles: without the V hand-over they race with the interior particle pass
    ("no V hand-over", lambda p: _edit(lambda op: (op[0], op[1], op[2], None if op[3] == "V" else op[3], op[4], op[5], op[6]))(_drop(lambda op: op[4] == "V" or op[0].startswith("signal V"))(p))),
    # the interior particles add up halo-side partial sums too: without `wait G` they read them before they exist
    ("no G wait", _drop(lambda op: op[4] == "G")),
    # interior tiles that touched a boundary particle would race with the boundary-particle pass on the halo queue
    ("interior tiles read boundary particles", _edit(lambda op: (op[0], dict(op[1], **{PRED_B: int(op[0][6:-1])}), op[2], op[3], op[4], op[5], op[6]) if op[0].startswith("T_int") else op)),
    # a signal kernel at the end of a call is what lets the halo queue's last wait finish
    ("no flush at the end of a call", _drop(lambda op: op[0].startswith("signal V"))),
    # the re-prediction rewrites the boundary particles the halo queue finished -- and its last transfer may still be reading them
    ("re-prediction not behind the halo queue", _edit(lambda op: op[:6] + (None,) if op[0] == "repredict" else op)),
    # the refresh exchange must not start before the new predictions exist
    ("refresh exchange not behind the re-prediction", _drop(lambda op: op[0] in ("signal D", "wait D"))),
])
def test_the_model_notices_a_broken_choreography(what, mutate):
    with pytest.raises(Violation):
        for calls in ([2], [3], [2, 2], [2, -2]):
            explore(calls, mutate)
