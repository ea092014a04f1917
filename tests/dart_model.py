"""Executable model of the PARALLEL contour formulation used by the HIP path (csrc/k_contours.hip).

Test infrastructure: plain Python, small images only.  It states, without any sequential
raster scan or label image, what the kernels compute, so that the design can be checked
against the oracle's sequential Suzuki-Abe restatement (oracle/a3_oracle.c,
a3o_find_contours) before/independently of the GPU:

  * a "dart" is (pixel p, direction k_in of a foreground 8-neighbour).  Its successor is
    (p + dir(k_out), opposite(k_out)) where k_out is the next foreground neighbour
    counter-clockwise after k_in.  succ is a bijection, so darts decompose into cycles;
    every border the reference traces is one of these cycles, rotated to its start dart.
  * start events: pixel q has a W-event if x>0 and its west neighbour is background, an
    E-event if x+1<W and its east neighbour is background; the event's dart is
    (q, first foreground neighbour clockwise from W resp. E).
  * which event starts a cycle is the unique fixpoint of
        T(c) = min{ key(e) : e event of c, fires(e; T) },
        Wfires(q)  = all cycles through q have T >= 2q,
        W-event(q) fires iff Wfires(q);  E-event(q) fires iff not (hasW(q) and Wfires(q)),
    with key = 2*raster(q) (+1 for E).  Contours are emitted in increasing T.
"""
import numpy as np

DX = [-1, -1, 0, 1, 1, 1, 0, -1]  # W NW N NE E SE S SW (clockwise on screen)
DY = [0, -1, -1, -1, 0, 1, 1, 1]
INF = 1 << 62


def pdart_mask(F: int) -> int:
    """Directions k whose dart (p,k) can lie on a real border: neighbour k foreground, neighbour k-1 background
    (the counter-clockwise sweep is not empty) and, for a 4-neighbour direction (k even), neighbour k-2 background
    too -- otherwise the dart only belongs to a triangular face cycle of the 8-neighbour graph."""
    rot1 = ((F << 1) | (F >> 7)) & 0xFF
    rot2 = ((F << 2) | (F >> 6)) & 0xFF
    return F & ~rot1 & (0xAA | ~rot2) & 0xFF


def contours_by_darts(binary: np.ndarray, max_iter: int = 64, node_rule: str = "any8"):
    """-> (list of contours as lists of (x, y), stats dict), same order/rotation as find_contours."""
    h, w = binary.shape
    fg = binary > 0

    def isfg(x, y):
        return 0 <= x < w and 0 <= y < h and fg[y, x]

    # neighbour masks
    F = np.zeros((h, w), dtype=np.int32)
    for y in range(h):
        for x in range(w):
            if fg[y, x]:
                m = 0
                for k in range(8):
                    if isfg(x + DX[k], y + DY[k]):
                        m |= 1 << k
                F[y, x] = m

    def is_node(x, y):
        if not fg[y, x] or F[y, x] == 0:
            return False
        if node_rule == "any8":
            return F[y, x] != 0xFF
        if node_rule == "pdart":
            return pdart_mask(int(F[y, x])) != 0
        # "border4": at least one background / out-of-image 4-neighbour
        return (F[y, x] & 0b01010101) != 0b01010101

    darts = {}  # (x,y,k) -> idx
    dart_list = []
    for y in range(h):
        for x in range(w):
            if is_node(x, y):
                dmask = pdart_mask(int(F[y, x])) if node_rule == "pdart" else int(F[y, x])
                for k in range(8):
                    if dmask >> k & 1:
                        darts[(x, y, k)] = len(dart_list)
                        dart_list.append((x, y, k))
    n = len(dart_list)
    succ = np.full(n, -1, dtype=np.int64)
    broken = 0
    for i, (x, y, k) in enumerate(dart_list):
        m = F[y, x]
        ko = k
        for s in range(1, 9):
            kk = (k - s) & 7
            if m >> kk & 1:
                ko = kk
                break
        nx, ny = x + DX[ko], y + DY[ko]
        j = darts.get((nx, ny, (ko + 4) & 7))
        if j is None:
            succ[i] = i  # successor lies on an interior pixel: only spurious face cycles do this
            broken += 1
        else:
            succ[i] = j

    # events
    ev_w = {}  # pixel raster q -> dart idx
    ev_e = {}
    for y in range(h):
        for x in range(w):
            if not fg[y, x] or F[y, x] == 0:
                continue
            q = y * w + x
            m = F[y, x]
            if x > 0 and not (m >> 0 & 1):
                for s in range(1, 8):
                    if m >> s & 1:
                        ev_w[q] = darts[(x, y, s)]
                        break
            if x + 1 < w and not (m >> 4 & 1):
                for s in range(1, 8):
                    kk = (4 + s) & 7
                    if m >> kk & 1:
                        ev_e[q] = darts[(x, y, kk)]
                        break

    # cycles (the kernels do this with pointer doubling; here: plain walks)
    cyc = np.full(n, -1, dtype=np.int64)
    cycles = []
    chain_has_event = 0
    event_darts = set(ev_w.values()) | set(ev_e.values())
    for i in range(n):
        if cyc[i] >= 0:
            continue
        # walk until we come back or hit a terminal
        path = []
        j = i
        seen_local = {}
        while cyc[j] < 0 and j not in seen_local and succ[j] != j:
            seen_local[j] = len(path)
            path.append(j)
            j = succ[j]
        if succ[j] == j and cyc[j] < 0:
            path.append(j)
            cid = len(cycles)
            cycles.append(None)  # broken chain, never traced
            for d in path:
                cyc[d] = cid
                if d in event_darts:
                    chain_has_event += 1
            continue
        if cyc[j] >= 0:
            cid = cyc[j]
            for d in path:
                cyc[d] = cid
                if cycles[cid] is None and d in event_darts:
                    chain_has_event += 1
            continue
        start = seen_local[j]
        assert start == 0, "succ must be a bijection on intact cycles"
        cid = len(cycles)
        cycles.append(path)
        for d in path:
            cyc[d] = cid

    # darts per pixel
    pix_darts = {}
    for i, (x, y, k) in enumerate(dart_list):
        pix_darts.setdefault(y * w + x, []).append(i)

    events_of = {}
    for q, d in ev_w.items():
        events_of.setdefault(cyc[d], []).append((2 * q, d, q))
    for q, d in ev_e.items():
        events_of.setdefault(cyc[d], []).append((2 * q + 1, d, q))

    T = {c: min(ev)[0] for c, ev in events_of.items()}
    S = {c: min(ev)[1] for c, ev in events_of.items()}
    T_natural = dict(T)
    # static_fire (csrc/k_contours.hip): does a cycle's SMALLEST event fire whatever the other cycles do?  A W-event on a pixel that owns
    # this one dart only; an E-event on a pixel without a W side.  Such a cycle starts at its natural start under every assignment --
    # which is what lets k_local_contract finish with a short border without looking at its neighbours (kDead).
    # Round 5's extension (k_local_contract, E-events on pixels with a W side): the E-event fires iff the pixel's W-event does not,
    # and that cannot fire when a WITNESS exists -- a border through the pixel whose smallest event lies before the W-event and passes
    # the static test itself (it starts there under every assignment).
    static_ok, static_ok_wide = {}, {}
    for c, evs in events_of.items():
        key, d, q = min(evs)
        static_ok[c] = (len(pix_darts[q]) == 1) if not (key & 1) else (q not in ev_w)
    for c, evs in events_of.items():
        key, d, q = min(evs)
        static_ok_wide[c] = static_ok[c] or bool((key & 1) and any(T_natural.get(cyc[e], INF) < 2 * q and static_ok.get(cyc[e], False)
                                                                    for e in pix_darts[q]))
    # does every border's smallest event fire under the NATURAL assignment (each border starts at its smallest event)?  If so that
    # assignment is the fixpoint -- what k_cycle_select's inline check and k_local_contract's trust_natural rule rely on.
    natural_fires = {}
    for c, evs in events_of.items():
        key, d, q = min(evs)
        wf = all(T_natural.get(cyc[e], INF) >= 2 * q for e in pix_darts[q])
        natural_fires[c] = (not (q in ev_w and wf)) if (key & 1) else wf
    iters = 0
    while True:
        iters += 1

        def wfires(q):
            return all(T.get(cyc[d], INF) >= 2 * q for d in pix_darts[q])

        T2, S2 = {}, {}
        for c, evs in events_of.items():
            best = (INF, -1)
            for key, d, q in evs:
                if key & 1:
                    fires = not (q in ev_w and wfires(q))
                else:
                    fires = wfires(q)
                if fires and key < best[0]:
                    best = (key, d)
            if best[0] < INF:
                T2[c], S2[c] = best
        if T2 == T and S2 == S:
            break
        T, S = T2, S2
        if iters >= max_iter:
            raise RuntimeError("start resolution did not converge")

    out = []
    for c in sorted(T, key=lambda c: T[c]):
        path = cycles[c]
        assert path is not None, "an event dart on a broken chain"
        s = path.index(S[c])
        rot = path[s:] + path[:s]
        out.append([(dart_list[d][0], dart_list[d][1]) for d in rot])
    stats = {"darts": n, "broken": broken, "cycles": len(cycles), "iters": iters, "chain_events": chain_has_event,
             "T_final": dict(T), "T_natural": T_natural, "natural_fires": natural_fires, "static_ok": static_ok, "static_ok_wide": static_ok_wide, "cycle_len": {c: len(cycles[c]) for c in events_of if cycles[c] is not None}}
    return out, stats
