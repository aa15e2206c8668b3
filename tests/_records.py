"""Pure-Python restatement of the packed-record semantics (sgcount_amd/csrc/sgc_format.h),
used to check the host packer on CPU.  Test infrastructure only."""

CODE = {ord("A"): 0, ord("C"): 1, ord("G"): 2, ord("T"): 3}


def revcomp(seq: bytes) -> bytes:
    # fxread's complement as restated in the oracle: c & 2 ? c ^ 4 : c ^ 21 over the reversed bytes
    return bytes((c ^ 4) if (c & 2) else (c ^ 21) for c in reversed(seq))


def expected_windows(seq: bytes, L: int, reverse: bool, o: int, recursion: bool):
    """→ {"C"|"P"|"M": (state, key)}; state 0 clean, 1 dead, 2+j one 'N' at j (counter.rs:96-180)."""
    n = len(seq)
    s = revcomp(seq) if reverse else seq
    out = {}
    alive = True
    for name, p in (("C", o), ("P", o + 1), ("M", o - 1)):
        if name != "C" and not recursion:
            alive = False
        w = s[p:p + L] if (alive and p >= 0 and p + L <= n) else None
        if w is None:
            alive = False            # bounds failure ends the chain (counter.rs:105-108)
            out[name] = (1, None)
            continue
        inv = [(j, c) for j, c in enumerate(w) if c not in CODE]
        key = 0
        for j, c in enumerate(w):
            key |= CODE.get(c, 0) << (2 * j)
        if not inv:
            out[name] = (0, key)
        elif len(inv) == 1 and inv[0][1] == ord("N"):
            out[name] = (2 + inv[0][0], key)
        else:
            out[name] = (1, None)
    return out


def decode_record(words, L):
    """words: tuple of 1 (rec8) or 2 (rec16) ints → {"C","P","M": (state, key)}"""
    K = L + 2
    if len(words) == 1:
        span = words[0] & ((1 << (2 * K)) - 1)
        status = words[0] >> (2 * K)
    else:
        span, status = words
    sC, sP, sM = status % K, (status // K) % K, status // (K * K)
    mask = (1 << (2 * L)) - 1
    keys = {"C": (span >> 2) & mask, "P": (span >> 4) & mask, "M": span & mask}
    st = {"C": sC, "P": sP, "M": sM}
    return {k: (st[k], None if st[k] == 1 else keys[k]) for k in st}
