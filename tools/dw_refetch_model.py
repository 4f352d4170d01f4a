"""What the grouped weight-gradient launch MUST fetch because the chip has eight separate L2s -- host-only arithmetic, no GPU.

Replays the launch's own placement (csrc/gemm_bf16.hip grouped_launch: tile kind per problem from the planner, problems ordered longest workgroup first,
XCD runs cut per tile-shape class, supertile order inside a problem) and counts, per XCD, the operand panels its tiles touch: an XCD's L2 must take
every panel (BM x K of X, BN x K of dY) at least once however well its tiles are ordered and synchronised.  The sum over the eight XCDs is the launch's
fetch floor under the present partition; the same sum with every panel counted once is the algorithmic read.  If the measured FETCH_SIZE sits at the
floor, the re-fetch is the PARTITION (eight L2s each needing the panels of the tiles they own), not the order or the timing of the tiles.

python tools/dw_refetch_model.py [cfg3|cfg4|cfg2]      (prints the table that went into profiles/r05_dw_refetch.txt)"""
import itertools, math, sys

CFG = {"cfg2": (4096, 64, 10), "cfg3": (16384, 128, 10), "cfg4": (8192, 256, 50)}
pad = lambda n: (n + 63) // 64 * 64


def problems(D, K):
    Dp, Kp, H, T = pad(D), pad(K), 2048, 512
    return [("enc0", 832, T), ("enc1", T, T), ("zc", T, 2 * H), ("mv", H, 2 * Dp), ("lg", H, Kp),
            ("dec0", Dp, H), ("dec1", H, 512), ("dec2", 512, 512), ("out", 512, 832)]


def best_kind(M, N): return 0 if (M % 128 == 0 and N % 128 == 0) else (1 if M % 128 == 0 else 2)
def dims(kind): return (64 if kind == 2 else 128), (128 if kind == 0 else 64)
def ntiles(M, N, kind): bm, bn = dims(kind); return (M // bm) * (N // bn)
def wg_bytes(kind, Kc): bm, bn = dims(kind); return 2.0 * (bm + bn) * Kc


def plan(probs, Kc):
    """grouped_launch's tile plan: least-loaded busiest CU (workgroup j -> CU j mod 256, longest first), ties to the fewest bytes"""
    lo = [best_kind(M, N) for _, M, N in probs]
    best, best_max, best_sum = None, 1e300, 1e300
    for cur in itertools.product(*[range(k, 3) for k in lo]):
        s = []
        for (_, M, N), k in zip(probs, cur): s += [wg_bytes(k, Kc)] * ntiles(M, N, k)
        s.sort(reverse=True)
        load = [0.0] * 256
        for j, b in enumerate(s): load[j & 255] += b
        mx, sm = max(load), sum(s)
        if mx < best_max * 0.99 or (mx < best_max * 1.01 and sm < best_sum):
            best_max, best_sum, best = min(mx, best_max), sm, cur
    return list(best)


def auto_group_m(tm, tn, bm, bn, run):
    gm = int(math.sqrt(run * bn / bm) + 0.5)
    need = int((run + tn - 1) // tn)
    return max(1, min(max(gm, need), tm))


def xcd_runs(n):
    """xcd_run_index over [0, n): XCD x owns the contiguous items [run0, run0 + count)"""
    out, r0 = [], 0
    for x in range(8):
        cnt = (n - 1 - x) // 8 + 1 if x < n else 0
        out.append((r0, r0 + cnt)); r0 += cnt
    return out


def model(name):
    B, D, K = CFG[name]
    probs = problems(D, K)
    kinds = plan(probs, B)
    order = sorted(range(len(probs)), key=lambda i: -wg_bytes(kinds[i], B))          # stable, longest first
    seq = [(probs[i], kinds[i]) for i in order]
    starts, tot = [], 0
    for (nm, M, N), k in seq: starts.append(tot); tot += ntiles(M, N, k)
    starts.append(tot)
    # classes: runs of consecutive problems of one tile shape (K is the same for all here)
    cls, lo = [], 0
    while lo < len(seq):
        hi = lo + 1
        while hi < len(seq) and seq[hi][1] == seq[lo][1]: hi += 1
        cls.append((lo, hi)); lo = hi
    per_xcd = [dict() for _ in range(8)]          # (problem, 'A'|'B', panel) -> bytes
    uniq = {}
    for lo, hi in cls:
        c0, c1 = starts[lo], starts[hi]
        run = max(1.0, (c1 - c0) / 8.0)
        runs = xcd_runs(c1 - c0)
        for x, (a, b) in enumerate(runs):
            for item in range(c0 + a, c0 + b):
                i = max(j for j in range(lo, hi) if starts[j] <= item)
                (nm, M, N), k = seq[i]
                bm, bn = dims(k); tm_, tn_ = M // bm, N // bn
                gm_max = auto_group_m(tm_, tn_, bm, bn, min(float(tm_ * tn_), run))
                bid = item - starts[i]
                gsz = gm_max * tn_; grp = bid // gsz; first = grp * gm_max
                gm = min(tm_ - first, gm_max); inn = bid - grp * gsz
                tm, tn = first + inn % gm, inn // gm
                for key, byt in (((nm, "A", tm), 2.0 * bm * B), ((nm, "B", tn), 2.0 * bn * B)):
                    per_xcd[x][key] = byt; uniq[key] = byt
    floor = sum(sum(d.values()) for d in per_xcd)
    alg = sum(uniq.values())
    pmv = sum(M * N for _, M, N in probs) * 12.0
    print("%s: %d rows, tile kinds %s (0 = 128x128, 1 = 128x64, 2 = 64x64), %d workgroups" % (name, B, dict((probs[i][0], kinds[i]) for i in range(len(probs))), tot))
    print("  operands, each panel once (algorithmic): %7.1f MB      + parameters / m / v read by the fused update: %5.1f MB" % (alg / 1e6, pmv / 1e6))
    print("  operands, each panel once PER XCD that owns a tile of it (floor of this partition): %7.1f MB = %.2f x" % (floor / 1e6, floor / alg))
    for (nm, M, N) in probs:
        a = sum(v for d in per_xcd for k, v in d.items() if k[0] == nm); u = sum(v for k, v in uniq.items() if k[0] == nm)
        nx = sum(1 for d in per_xcd if any(k[0] == nm for k in d))
        print("    %-5s %4d x %4d: %6.1f MB once, %6.1f MB over the %d XCDs that own its tiles (%.2f x)" % (nm, M, N, u / 1e6, a / 1e6, nx, a / u))
    print("  fetch floor incl. p / m / v: %.1f MB = %.2f x of %.1f MB" % ((floor + pmv) / 1e6, (floor + pmv) / (alg + pmv), (alg + pmv) / 1e6))
    return floor, alg, pmv


if __name__ == "__main__":
    for n in (sys.argv[1:] or ["cfg4", "cfg3", "cfg2"]):
        model(n)


def model_cost_cuts(name):
    """the same launch with the XCD runs cut over the WHOLE problem sequence by streamed bytes instead of per tile-shape class by count"""
    B, D, K = CFG[name]
    probs = problems(D, K)
    kinds = plan(probs, B)
    order = sorted(range(len(probs)), key=lambda i: -wg_bytes(kinds[i], B))
    seq = [(probs[i], kinds[i]) for i in order]
    items = []           # (problem index in seq, bid)
    for i, ((nm, M, N), k) in enumerate(seq): items += [(i, b) for b in range(ntiles(M, N, k))]
    cost = [wg_bytes(seq[i][1], B) for i, _ in items]
    total = sum(cost)
    cuts, acc, x = [0], 0.0, 1
    for j, c in enumerate(cost):
        acc += c
        while x < 8 and acc >= total * x / 8.0 - 1e-9: cuts.append(j + 1); x += 1
    while len(cuts) < 9: cuts.append(len(items))
    cuts[8] = len(items)
    per_xcd = [dict() for _ in range(8)]; uniq = {}
    for x in range(8):
        run = max(1.0, float(cuts[x + 1] - cuts[x]))
        for i, bid in items[cuts[x]:cuts[x + 1]]:
            (nm, M, N), k = seq[i]
            bm, bn = dims(k); tm_, tn_ = M // bm, N // bn
            gm_max = auto_group_m(tm_, tn_, bm, bn, min(float(tm_ * tn_), run))
            gsz = gm_max * tn_; grp = bid // gsz; first = grp * gm_max
            gm = min(tm_ - first, gm_max); inn = bid - grp * gsz
            tm, tn = first + inn % gm, inn // gm
            for key, byt in (((nm, "A", tm), 2.0 * bm * B), ((nm, "B", tn), 2.0 * bn * B)):
                per_xcd[x][key] = byt; uniq[key] = byt
    floor = sum(sum(d.values()) for d in per_xcd); alg = sum(uniq.values())
    print("%s, XCD runs cut by streamed bytes over the whole sequence: tiles per XCD %s; floor %.1f MB = %.2f x" %
          (name, [cuts[x + 1] - cuts[x] for x in range(8)], floor / 1e6, floor / alg))
    for (nm, M, N) in probs:
        a = sum(v for d in per_xcd for k, v in d.items() if k[0] == nm); u = sum(v for k, v in uniq.items() if k[0] == nm)
        nx = sum(1 for d in per_xcd if any(k[0] == nm for k in d))
        print("    %-5s %6.1f MB once, %6.1f MB over %d XCDs (%.2f x)" % (nm, u / 1e6, a / 1e6, nx, a / u))


if __name__ == "__main__" and "--cuts" in sys.argv:
    pass
