"""TEST INFRASTRUCTURE (not part of the product): a restatement of the reference's large-insertion report,
SortOutputLI (src/reporter.cpp:1853-2141, called from SearchSVs, src/pindel.cpp:1167-1169), which is the ONE consumer
in the reference of the reads that have a close end and no far end.  The product declares LI out of scope (DESIGN.md
section 10); here it only extends the golden pin of the search from the 10 968 reads that reach _D/_SI/_TD/_INV to the
3 894 that do not: their UP_Close.back() (AbsLoc, LengthStr) and their flipped sequences decide every byte of _LI.

What it takes from elsewhere, and why that is sound:
  * "Used || !UP_Far.empty()" (reporter.cpp:1883, 1953): the classifiers only ever set Used on reads WITH a far end
    (searchdeletions.cpp / search_variant.cpp skip `UP_Far.empty()`), so the test is `UP_Far.empty()`;
  * CurrentChrMask: a 'B' at the breakpoints of every event the _D/_SI/_TD/_INV reporters wrote
    (reporter.cpp:194-195, 330-336, 517-518, 676-679, 802-803, 1625-1626).  Each is recoverable from the event header the
    same function prints (masked_positions below), so the mask comes from the report files -- the reference-held gold
    ones, which both routes reproduce byte for byte (tests/test_golden_pin.py).
"""
import re

SPACER = 100000
MAX_SHORT = 128           # pindel.h:126
_RC = bytes.maketrans(b"ACGTN", b"TGCAN")


def reverse_complement(s: bytes) -> bytes:
    """ReverseComplement (Convert2RC4N, pindel.cpp:966-970): anything but ACGTN becomes \\0."""
    return bytes(_RC[c] if c in b"ACGTN" else 0 for c in s[::-1])


def cap2low(s: bytes) -> bytes:
    """Cap2Low (Cap2LowArray, pindel.cpp:971-976: A C G T N and '$'; every other entry of the zeroed table is \\0)"""
    t = {ord("A"): ord("a"), ord("C"): ord("c"), ord("G"): ord("g"), ord("T"): ord("t"), ord("N"): ord("n"), ord("$"): ord("n")}
    return bytes(t.get(c, 0) for c in s)


def masked_positions(reports: dict) -> set:
    """AbsLoc of every 'B' the four reporters leave in CurrentChrMask, from the event headers they print.
    reports: suffix -> bytes of the report file."""
    out = set()
    for suffix, data in reports.items():
        for line in data.split(b"\n"):
            if b"\tSupports " not in line or b"\tBP " not in line:
                continue
            f = line.split(b"\t")
            bp = f.index(next(x for x in f if x.startswith(b"BP ")))
            a, b = int(f[bp].split()[1]), int(f[bp + 1])
            rg = next(i for i, x in enumerate(f) if x.startswith(b"BP_range "))
            c, d = int(f[rg].split()[1]), int(f[rg + 1])
            kind = f[1].split()[0]
            nt = next(x for x in f if x.startswith(b"NT "))
            if suffix == "TD" or (kind == b"INV" and re.match(rb"NT \d+:\d+ ", nt)):
                # header shows BPLeft, BPRight + 2 (reporter.cpp:202, 524)
                out.update((a + SPACER, b - 2 + SPACER))
            else:
                # header shows BPLeft + 1, BPRight + 1 and (D / SI) RealStart + 1, RealEnd + 1 (reporter.cpp:344, 686, 808, 1632)
                out.update((a - 1 + SPACER, b - 1 + SPACER))
                if suffix in ("D", "SI"):
                    out.update((c - 1 + SPACER, d - 1 + SPACER))
    return out


class LIRead:
    __slots__ = ("name", "seq", "strand", "pos", "ms", "tag", "frag", "close_abs", "close_len", "has_far", "length")


def sort_output_li(chr_seq: bytes, reads, mask: set, window_start: int, window_end: int, max_insert_size: int,
                   report_length: int, samples, cutoff: int = 1, count_start: int = 0) -> bytes:
    """SortOutputLI for one window.  reads: LIRead in Reads_SR order (only reads WITH a close end).  Returns the text
    the reference appends to <prefix>_LI."""
    border = 4 * max_insert_size
    abs_start = SPACER + window_start
    abs_end = min(SPACER + window_end, len(chr_seq) - SPACER)
    lo, hi = abs_start - border, abs_end + border            # ShiftedVector(start, end): indices clamp to [lo, hi]

    def clamp(p):
        return min(max(p, lo), hi)

    plus, minus, event = {}, {}, {}
    for r in reads:                                              # reporter.cpp:1882-1897
        if r.has_far:
            continue
        p = clamp(r.close_abs)
        if r.strand == "+" and plus.get(p, 0) < MAX_SHORT:
            plus[p] = plus.get(p, 0) + 1
        if r.strand == "-" and minus.get(p, 0) < MAX_SHORT:
            minus[p] = minus.get(p, 0) + 1
    # candidate (plus, minus) position pairs, reporter.cpp:1909-1943 (the loops modify Index_Minus as they go)
    positions = []
    im = lo
    while im < hi:
        skip_plus = False
        for m in range(im + 10, im - 11, -1):
            if m in mask:
                im = m + 10
                skip_plus = True
                break
        if not skip_plus and minus.get(clamp(im), 0) >= cutoff:
            ip = im - 1
            while ip <= im + 30:                                 # (the bound is re-read every iteration: im may move)
                skip_this = False
                for m in range(ip + 10, ip - 11, -1):
                    if m in mask:
                        if m + 10 > im:
                            im = m + 10
                        skip_this = True
                        break
                if not skip_this and plus.get(clamp(ip), 0) >= cutoff:
                    positions.append({"plus": ip, "minus": im, "P": [], "M": []})
                    event[clamp(ip)] = len(positions) - 1
                    event[clamp(im)] = len(positions) - 1
                ip += 1
        im += 1
    for idx, r in enumerate(reads):                              # reporter.cpp:1952-1969
        if r.has_far:
            continue
        e = event.get(clamp(r.close_abs), -1)
        if e == -1:
            continue
        positions[e]["P" if r.strand == "+" else "M"].append(idx)
    out = []
    count = count_start
    for pos in positions:                                        # reporter.cpp:1982-2134
        if not pos["M"] or not pos["P"]:
            continue
        pr, mr = [reads[i] for i in pos["P"]], [reads[i] for i in pos["M"]]
        bal = set()
        for tag, group in (("M", mr), ("P", pr)):
            for r in group:
                half = r.length * 0.5
                if float(r.close_len) > half:
                    bal.add(tag + "+")
                elif float(r.close_len) < half:
                    bal.add(tag + "-")
        sup_p = {s: 0 for s in samples}
        sup_m = {s: 0 for s in samples}
        for r in mr:
            sup_m[r.tag] += 1
        for r in pr:
            sup_p[r.tag] += 1
        if not any(sup_p[s] > 0 and sup_m[s] > 0 for s in samples) or len(bal) < 1:
            continue
        pp, mp = pos["plus"], pos["minus"]
        out.append(b"#" * 56)
        head = f"{count}\tLI\tChrID {pr[0].frag}\t{pp - SPACER + 1}\t+ {len(pr)}\t{mp - SPACER + 1}\t- {len(mr)}"
        for s in samples:                                        # indexToSampleMap: sample names in sorted (std::map) order
            head += f"\t{s} + {sup_p[s]} - {sup_m[s]}"
        count += 1
        out.append(head.encode())
        out.append(chr_seq[pp - report_length + 1:pp + 1] + cap2low(chr_seq[pp + 1:pp + 1 + report_length]))
        for r in pr:
            line = b" " * max(0, report_length - r.close_len) + reverse_complement(r.seq)
            line += f"\t{r.strand}\t{r.pos}\t{r.ms}\t{r.tag}\t{r.name}".encode()
            out.append(line)
        out.append(b"-" * 56)
        out.append(cap2low(chr_seq[mp - report_length:mp]) + chr_seq[mp:mp + report_length])
        for r in mr:
            line = b" " * max(0, report_length + r.close_len - r.length) + r.seq
            line += f"{r.strand}\t{r.pos}\t{r.ms}\t{r.tag}\t{r.name}".encode()       # (no tab before MatchedD: reporter.cpp:2124)
            out.append(line)
    return b"\n".join(out) + (b"\n" if out else b"")
