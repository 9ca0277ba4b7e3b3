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
PRED_G1 = "pred ghost (second buffer)"   # peer-to-peer halo: the ghosts of odd substeps
# two-layer ghost regions: the first ghost layer's local predictions, the second-layer ghost tets' state, and the receive buffers
# (even-substep data / the early, odd-substep data), two sets each
G1_LOCAL, L2_STATE, EVEN_SET, ODD_SET = "pred first layer (local)", "second-layer tet state", ("even buffers set 0", "even buffers set 1"), ("odd buffer set 0", "odd buffer set 1")
DEEP_BUFS = (G1_LOCAL, L2_STATE) + EVEN_SET + ODD_SET


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


def _words(x):
    return () if not x else (x,) if isinstance(x, str) else tuple(x)


PEER = "peer:"   # prefix of a buffer / word that lives at the OTHER rank (peer-to-peer halo: stores and raises into the neighbour's memory)


def explore(calls, mutate=None, program=None):
    """Depth-first search over all interleavings of 2 ranks x 2 queues.  Returns the number of distinct states visited."""
    program = program or rank_program
    progs = [program(calls), program(calls)]
    if mutate:
        progs = [mutate(p) for p in progs]
    queues = [progs[0][0], progs[0][1], progs[1][0], progs[1][1]]   # rank = q // 2
    total = sum(abs(n) for n in calls)
    bufs = (PRED_B, PRED_I, PRED_G, PART_H, PART_I, PRED_G1) + DEEP_BUFS
    version0 = tuple((0,) * len(bufs) for _ in range(2))           # [rank][buffer]
    # state: (pc per queue, running flag per queue, words per rank (G, V, D, A0, A1), versions)
    start = ((0, 0, 0, 0), (False,) * 4, ((0,) * 9, (0,) * 9), version0)
    seen, stack = {start}, [start]
    bi = {b: i for i, b in enumerate(bufs)}
    wi = {"G": 0, "V": 1, "D": 2, "A0": 3, "A1": 4, "E0": 5, "E1": 6, "O0": 7, "O1": 8}
    p2p = program is not rank_program
    deep = getattr(program, "deep", False)

    def where(r, name):   # (rank, plain name) of a buffer or word that may carry the peer prefix
        return (1 - r, name[len(PEER):]) if name.startswith(PEER) else (r, name)

    def active(state):   # (queue, op) of every kernel between START and END
        pcs, running = state[0], state[1]
        return [(q, queues[q][pcs[q]]) for q in range(4) if running[q]]

    while stack:
        state = stack.pop()
        pcs, running, words, versions = state
        if all(pcs[q] == len(queues[q]) for q in range(4)):
            if any(w for r in words for w in r):
                raise Violation("a word is still raised at the end: %r" % (words,))
            final_ghosts = (PRED_G1 if total & 1 else PRED_G) if p2p else PRED_G   # peer-to-peer: the buffer of the next substep's parity
            check = (PRED_B, PRED_I) if deep else (PRED_B, PRED_I, final_ghosts)
            if any(versions[r][bi[b]] != total for r in range(2) for b in check):
                raise Violation("final versions %r" % (versions,))
            continue
        moves = []
        for q in range(4):
            if pcs[q] == len(queues[q]):
                continue
            name, reads, writes, raises, waits, xfer, join = queues[q][pcs[q]][:7]
            extra = queues[q][pcs[q]][7] if len(queues[q][pcs[q]]) > 7 else {}
            r = q // 2
            if not running[q]:
                # ---- START
                if waits is not None and not words[r][wi[waits]]:
                    continue                                   # the wait kernel spins: model it as "cannot complete yet"
                if any(not words[r][wi[wd]] for wd in _words(extra.get("await"))):
                    continue                                   # a kernel whose waves look at the word(s) themselves: no wave gets past its first instructions
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
            name, reads, writes, raises, waits, xfer, join = queues[q][pcs[q]][:7]
            extra = queues[q][pcs[q]][7] if len(queues[q][pcs[q]]) > 7 else {}
            r = q // 2
            npcs, nrun, nwords, nver = list(pcs), list(running), [list(w) for w in words], [list(v) for v in versions]
            if kind == "start":
                if xfer is not None:                           # both directions at once; atomic (START and END together)
                    for rr in (0, 1):
                        if versions[rr][bi[PRED_B]] != xfer[2]:
                            raise Violation("%s sends boundary predictions of version %d" % (name, versions[rr][bi[PRED_B]]))
                    gbuf = xfer[3] if len(xfer) > 3 else PRED_G   # (peer-to-peer bodies: the refresh lands in the buffer of the current parity)
                    for rr in (0, 1):                          # the receiver's ghosts and the sender's boundary particles must not be in use
                        for aq, op in active(state):
                            if aq // 2 == rr and (gbuf in op[1] or PRED_B in op[2]):
                                raise Violation("%s while %s is running on rank %d" % (name, op[0], rr))
                            if aq // 2 != rr and (PEER + gbuf) in op[2]:
                                raise Violation("%s while %s is storing into the same ghosts" % (name, op[0]))
                        nver[rr][bi[gbuf]] = xfer[2]
                    npcs[q] += 1
                    npcs[(1 - r) * 2 + 1] += 1
                else:
                    if waits is not None:
                        nwords[r][wi[waits]] = 0               # consumed (wait kernels are modelled as atomic)
                        npcs[q] += 1
                    else:
                        for wd in _words(extra.get("clear")):  # put back a word whose waiters were the waves of the kernel in front
                            if not nwords[r][wi[wd]]:
                                raise Violation("%s clears %s, which is not raised" % (name, wd))
                            nwords[r][wi[wd]] = 0
                        for word in ((raises,) if isinstance(raises, str) else (raises or ())):
                            wr, wn = where(r, word)
                            if nwords[wr][wi[wn]]:
                                raise Violation("%s raises %s twice" % (name, word))
                            nwords[wr][wi[wn]] = 1
                        for b, want in reads.items():
                            if versions[r][bi[b]] != want:
                                raise Violation("%s reads %s at version %d, wants %d" % (name, b, versions[r][bi[b]], want))
                        # nobody may be writing what we read or write, or reading what we write -- buffers named per rank, so a
                        # store into the peer's memory is checked against the PEER's kernels
                        mine_r = {where(r, b) for b in reads}
                        mine_w = {where(r, b) for b in writes}
                        for aq, op in active(state):
                            ar = aq // 2
                            for b in op[2]:
                                if where(ar, b) in mine_r or where(ar, b) in mine_w:
                                    raise Violation("%s starts while %s writes %s" % (name, op[0], b))
                            for b in op[1]:
                                if where(ar, b) in mine_w:
                                    raise Violation("%s would write %s while %s reads it" % (name, b, op[0]))
                        if not reads and not writes:           # a signal kernel: atomic
                            npcs[q] += 1
                        else:
                            nrun[q] = True
            else:
                for b, new in writes.items():
                    br, bn = where(r, b)
                    nver[br][bi[bn]] = new
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


# ---- peer-to-peer halo (tetsim_halo_p2p_connect): no transfer; the boundary-particle kernel stores into the PEER's ghost buffer of the
# next substep's parity, the wait kernel in front of the halo-side tiles raises the peer's "arrived" word of that parity as it starts and
# waits for its own ------------------------------------------------------------------------------------------------------------------
def rank_program_p2p(calls, raise_in_own_kernel=False, fold=False, fold_halo=False):
    """As rank_program, for a connected body.  Substep s reads ghost buffer s & 1; P_b(s) also writes the peer's buffer (s + 1) & 1; the
    words A0 / A1 ("arrived", by parity) live at the receiver.  raise_in_own_kernel: partitions of one process give the raise a kernel of
    its own right behind P_b (tetsim_group_step_n), one rank per process folds it into the next wait kernel / the flush.
    fold (tetsim_halo.hip: interior_particles, the default of a connected body): no `wait G` kernel -- the interior particle kernel's
    waves look at G themselves (an 8th element {"await": word}) and nobody consumes it there; the main queue's NEXT operation -- the
    interior tiles of the next substep, or the flush's signal -- puts it back as it starts ({"clear": word}), in front of its raise.
    fold_halo (one rank per process): the halo queue's wait kernel is gone too -- the halo-side tiles look at V and at the neighbour's
    "arrived" word themselves, the boundary-particle kernel behind them puts both back as it starts, in front of its raise of G."""
    main, halo = [], []
    s = 0
    gb = lambda k: PRED_G1 if k & 1 else PRED_G
    pending = False   # P_b's "arrived" not raised yet
    for call, n in enumerate(calls):
        if n < 0:
            n = -n
            main.append(("repredict", {PRED_B: s, PRED_I: s}, {PRED_B: s, PRED_I: s}, None, None, None, len(halo)))
            main.append(("signal D", {}, {}, "D", None, None, None))
            halo.append(("wait D", {}, {}, None, "D", None, None))
            halo.append(("X(refresh %d)" % call, {PRED_B: s}, {}, None, None, ("refresh", call, s, gb(s)), None))   # RCCL / copies, into the current parity's buffer
        v_pending = False
        for _ in range(n):
            main.append(("T_int(%d)" % s, {PRED_I: s}, {PART_I: s + 1}, "V" if v_pending else None, None, None, None) + (({"clear": "G"},) if fold and v_pending else ()))
            # the wait kernel: raises (as it starts), then V, then the peer's "arrived" of this parity
            if pending:
                halo.append(("raise A%d" % (s & 1), {}, {}, PEER + "A%d" % (s & 1), None, None, None))
                pending = False
            looked = (["V"] if v_pending else []) + (["A%d" % (s & 1)] if s > 0 else [])
            if fold_halo:
                halo.append(("T_H(%d)" % s, {PRED_B: s, PRED_I: s, gb(s): s}, {PART_H: s + 1}, None, None, None, None, {"await": looked}))
                halo.append(("P_b(%d)" % s, {PART_H: s + 1}, {PRED_B: s + 1, PEER + gb(s + 1): s + 1}, "G", None, None, None, {"clear": looked}))
            else:
                if v_pending:
                    halo.append(("wait V(%d)" % (s - 1), {}, {}, None, "V", None, None))
                if s > 0:
                    halo.append(("wait A%d(%d)" % (s & 1, s), {}, {}, None, "A%d" % (s & 1), None, None))
                halo.append(("T_H(%d)" % s, {PRED_B: s, PRED_I: s, gb(s): s}, {PART_H: s + 1}, None, None, None, None))
                halo.append(("P_b(%d)" % s, {PART_H: s + 1}, {PRED_B: s + 1, PEER + gb(s + 1): s + 1}, "G", None, None, None))
            pending = True
            if raise_in_own_kernel:
                halo.append(("raise A%d" % ((s + 1) & 1), {}, {}, PEER + "A%d" % ((s + 1) & 1), None, None, None))
                pending = False
            if fold:
                main.append(("P_i(%d)" % s, {PART_H: s + 1, PART_I: s + 1}, {PRED_I: s + 1}, None, None, None, None, {"await": "G"}))
            else:
                main.append(("wait G(%d)" % s, {}, {}, None, "G", None, None))
                main.append(("P_i(%d)" % s, {PART_H: s + 1, PART_I: s + 1}, {PRED_I: s + 1}, None, None, None, None))
            v_pending = True
            s += 1
        main.append(("signal V(%d)" % (s - 1), {}, {}, "V", None, None, None) + (({"clear": "G"},) if fold else ()))
        if pending:
            halo.append(("raise A%d" % (s & 1), {}, {}, PEER + "A%d" % (s & 1), None, None, None))
            pending = False
        halo.append(("wait V(%d)" % (s - 1), {}, {}, None, "V", None, None))
    return main, halo


def _final_arrived_is_expected(fn):
    """The last substep's "arrived" stays raised at the end of a run (the next call's first wait consumes it): clear it in the model by
    appending that wait."""
    def program(calls):
        main, halo = fn(calls)
        total = sum(abs(n) for n in calls)
        return main, halo + [("wait A%d(next call)" % (total & 1), {}, {}, None, "A%d" % (total & 1), None, None)]
    return program


P2P = _final_arrived_is_expected(rank_program_p2p)
P2P_GROUP = _final_arrived_is_expected(lambda calls: rank_program_p2p(calls, raise_in_own_kernel=True))
P2P_FOLD = _final_arrived_is_expected(lambda calls: rank_program_p2p(calls, fold=True))
P2P_GROUP_FOLD = _final_arrived_is_expected(lambda calls: rank_program_p2p(calls, raise_in_own_kernel=True, fold=True))
P2P_FOLD_BOTH = _final_arrived_is_expected(lambda calls: rank_program_p2p(calls, fold=True, fold_halo=True))


@pytest.mark.parametrize("program", [P2P, P2P_GROUP, P2P_FOLD, P2P_GROUP_FOLD, P2P_FOLD_BOTH],
                         ids=["one rank per process", "ranks of one process", "one rank per process, G awaited by the particle kernel", "ranks of one process, G awaited by the particle kernel",
                              "one rank per process, no wait kernel on either queue"])
@pytest.mark.parametrize("calls", [[1], [2], [3], [1, 1], [2, 1, 2], [4], [2, -2], [1, -1, -3, 2], [3, 3]])
def test_peer_to_peer_halo_every_interleaving_is_live_and_race_free(calls, program):
    assert explore(calls, program=program) > 10 * sum(abs(n) for n in calls)


@pytest.mark.parametrize("what,mutate", [
    # ONE ghost buffer: a fast neighbour's boundary-particle kernel of substep s stores the ghosts of s + 1 while this rank's
    # halo-side tiles of substep s still read the ghosts of s
    ("no double buffering", _edit(lambda op: (op[0], {(PRED_G if b == PRED_G1 else b): v for b, v in op[1].items()},
                                              {(PEER + PRED_G if b == PEER + PRED_G1 else b): v for b, v in op[2].items()}) + op[3:])),
    # ONE word: substep s + 1's raise can land before this rank's wait kernel (still waiting for V) has consumed substep s's
    ("one arrived word instead of a pair", _edit(lambda op: op[:3] + (PEER + "A0" if op[3] in (PEER + "A0", PEER + "A1") else op[3], "A0" if op[4] in ("A0", "A1") else op[4]) + op[5:])),
    # the halo-side tiles must not start before the neighbour's stores are complete
    ("no wait for arrived", _drop(lambda op: op[4] in ("A0", "A1") and "next call" not in op[0])),
    # without the raise at the end of a call (nothing follows the last boundary-particle kernel) the neighbour's next call waits for ever
    ("no raise at the end of the last call", lambda p: (p[0], [op for i, op in enumerate(p[1]) if i != max(j for j, o in enumerate(p[1]) if o[0].startswith("raise A"))])),
])
def test_the_model_notices_a_broken_peer_to_peer_choreography(what, mutate):
    with pytest.raises(Violation):
        for calls in ([2], [3], [2, 2], [3, 3]):
            explore(calls, mutate, program=P2P)


@pytest.mark.parametrize("what,mutate", [
    # nobody puts G back: the next substep's particle kernel finds it still raised and adds up halo-side partial sums of the substep before
    ("G never put back", _edit(lambda op: op[:7] + ({k: v for k, v in op[7].items() if k != "clear"},) if len(op) > 7 else op)),
    # the particle kernel's waves do not look at G: they add up halo-side partial sums that may not exist yet
    ("particle kernel does not await G", _edit(lambda op: op[:7] + ({k: v for k, v in op[7].items() if k != "await"},) if len(op) > 7 else op)),
])
def test_the_model_notices_a_broken_folded_wait(what, mutate):
    for program in (P2P_FOLD, P2P_FOLD_BOTH):
        with pytest.raises(Violation):
            for calls in ([2], [3], [2, 2]):
                explore(calls, mutate, program=program)


# ---- two-layer ghost region on the peer-to-peer halo (TETSIM_FLAG_DEEP_GHOSTS): ghosts cross every other substep ---------------------
def rank_program_deep(calls, one_set=False, no_late_evolve_wait=False):
    """Substep r, exchange e = r // 2, buffer set st = e & 1.  Even r reads the set's even buffers (version r) and leaves the second-layer
    tet state and the local first-layer predictions at r + 1; odd r reads those predictions, stores the NEXT set's even buffers (r + 1)
    into the peer, and then evolves the second-layer tets after the fact from the odd buffer (version r) the peer stored after ITS even
    substep.  Words: E0 / E1 = "your even buffers of set 0 / 1 are full", O0 / O1 = the early message of set 0 / 1."""
    main, halo = [], []
    r = 0
    sel = (lambda st: 0) if one_set else (lambda st: st)
    pending = None   # an even substep's early-message raise, folded into the next kernel of the halo queue
    for n in calls:
        v_pending = False
        for _ in range(n):
            st = (r >> 1) & 1
            main.append(("T_int(%d)" % r, {PRED_I: r}, {PART_I: r + 1}, "V" if v_pending else None, None, None, None))
            if pending:
                halo.append(("raise " + pending, {}, {}, PEER + pending, None, None, None))
                pending = None
            if v_pending:
                halo.append(("wait V(%d)" % (r - 1), {}, {}, None, "V", None, None))
            if r % 2 == 0:
                if r > 0:
                    halo.append(("wait E%d(%d)" % (sel(st), r), {}, {}, None, "E%d" % sel(st), None, None))
                halo.append(("T_H+L2(%d)" % r, {PRED_B: r, PRED_I: r, EVEN_SET[sel(st)]: r, L2_STATE: r}, {PART_H: r + 1, L2_STATE: r + 1}, None, None, None, None))
                halo.append(("P_b(%d)" % r, {PART_H: r + 1}, {PRED_B: r + 1, PEER + ODD_SET[sel(st)]: r + 1}, "G", None, None, None))
                halo.append(("P_g1(%d)" % r, {PART_H: r + 1, EVEN_SET[sel(st)]: r}, {G1_LOCAL: r + 1}, None, None, None, None))
                pending = "O%d" % sel(st)
            else:
                halo.append(("T_H(%d)" % r, {PRED_B: r, PRED_I: r, G1_LOCAL: r}, {PART_H: r + 1}, None, None, None, None))
                halo.append(("P_b(%d)" % r, {PART_H: r + 1}, {PRED_B: r + 1, PEER + EVEN_SET[sel(st ^ 1)]: r + 1}, "G", None, None, None))
                halo.append(("raise E%d" % sel(st ^ 1), {}, {}, PEER + "E%d" % sel(st ^ 1), None, None, None))
                if not no_late_evolve_wait:
                    halo.append(("wait O%d(%d)" % (sel(st), r), {}, {}, None, "O%d" % sel(st), None, None))
                halo.append(("T_L2 late(%d)" % r, {G1_LOCAL: r, ODD_SET[sel(st)]: r, L2_STATE: r}, {L2_STATE: r + 1}, None, None, None, None))
            main.append(("wait G(%d)" % r, {}, {}, None, "G", None, None))
            main.append(("P_i(%d)" % r, {PART_H: r + 1, PART_I: r + 1}, {PRED_I: r + 1}, None, None, None, None))
            v_pending = True
            r += 1
        main.append(("signal V(%d)" % (r - 1), {}, {}, "V", None, None, None))
        if pending:
            halo.append(("raise " + pending, {}, {}, PEER + pending, None, None, None))
            pending = None
        halo.append(("wait V(%d)" % (r - 1), {}, {}, None, "V", None, None))
    # what stays raised at the end of a run is consumed by the next call: the early word after an even last substep, the even word after an odd one
    total = sum(calls)
    last_set = ((total - 1) >> 1) & 1
    halo.append(("wait O%d(next call)" % sel(last_set), {}, {}, None, "O%d" % sel(last_set), None, None) if total % 2 else
                ("wait E%d(next call)" % sel(last_set ^ 1), {}, {}, None, "E%d" % sel(last_set ^ 1), None, None))
    return main, halo


rank_program_deep.deep = True


@pytest.mark.parametrize("calls", [[1], [2], [3], [4], [5], [2, 2], [1, 2, 1], [3, 2], [6]])
def test_two_layer_ghost_choreography_every_interleaving_is_live_and_race_free(calls):
    assert explore(calls, program=rank_program_deep) > 10 * sum(calls)


def _variant(**kw):
    def program(calls):
        return rank_program_deep(calls, **kw)
    program.deep = True
    return program


@pytest.mark.parametrize("what,program", [
    # ONE set of buffers and words: a neighbour needs nothing from this rank for its odd substep, so it can store the next exchange's
    # even buffers while this rank's tiles of the current even substep still read them
    ("one set of buffers", _variant(one_set=True)),
    # the second-layer tets must not be evolved before the neighbour's early message is there
    ("late evolve without waiting for the early message", _variant(no_late_evolve_wait=True)),
])
def test_the_model_notices_a_broken_two_layer_choreography(what, program):
    with pytest.raises(Violation):
        for calls in ([2], [4], [5], [2, 2], [6]):
            explore(calls, program=program)
