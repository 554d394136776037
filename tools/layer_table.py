"""Per-layer roofline table from rocprofv3 output of tools/profile_serial.py.

    python tools/layer_table.py --ops gpurun_out/ops_c2.json --stats <kernel-trace db> [--pmc <db> ...] > profiles/r02_layer_table_c2.csv

Dispatches are matched to plan ops by order within a step (a step starts at the stem kernel; only the last `steps` steps of
the trace are used, i.e. the serial, one-batch-in-flight phase).  Per conv launch: average kernel duration, TFLOP/s and GB/s
from the ALGORITHMIC flops / bytes (SURVEY.md 8d accounting), fraction of the layer's own bound
max(flops / 2.5 PF, bytes / 8 TB/s), and -- when PMC passes are given -- the SQ counters averaged per launch
(SQ_VALU_MFMA_BUSY_CYCLES etc.; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (kernel duration x 2.4 GHz x 1024 SIMDs): the fraction of
SIMD-cycles in which the matrix pipe was busy).
"""
import argparse
import json
import sqlite3
import sys
from collections import defaultdict

MFMA_PEAK, HBM_PEAK = 2.5e15, 8.0e12
CONV_LIKE = ("ymi::conv", "spp_pool", "ymi::c3_fused", "ymi::c3_tile", "ymi::stem_body1")     # kernels that correspond 1:1 to plan ops of kind conv / pool (incl. the fused head)
STEM = ("conv_stem", "letterbox")     # first kernel of a step


def dispatches_from_stats(db):
    cur = sqlite3.connect(db).cursor()
    return [(n, int(s), int(e)) for n, s, e in cur.execute("select name, start, end from kernels order by start")]


def split_steps(disp, n_steps):
    starts = [i for i, d in enumerate(disp) if d[0].startswith("void ymi::letterbox") or "conv_stem" in d[0] or "stem_body1" in d[0]]
    # a letterbox launch directly followed by a stem is ONE step start
    clean = []
    for i in starts:
        if clean and i - clean[-1] <= 2 and "letterbox" in disp[clean[-1]][0]:
            continue
        clean.append(i)
    clean = clean[-n_steps:]
    out = []
    for k, i in enumerate(clean):
        j = clean[k + 1] if k + 1 < len(clean) else len(disp)
        out.append(disp[i:j])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ops", required=True)
    ap.add_argument("--stats", required=True)
    ap.add_argument("--pmc", nargs="*", default=[])
    a = ap.parse_args()
    meta = json.load(open(a.ops))
    ops = [o for o in meta["ops"] if o["kind"] in ("conv", "pool")]
    steps = split_steps(dispatches_from_stats(a.stats), meta["steps"])
    dur = defaultdict(list)
    names = {}
    other = defaultdict(list)
    for st in steps:
        k = 0
        for n, s, e in st:
            short = n.split("(")[0].replace("void ymi::", "")
            if any(t in n for t in CONV_LIKE) and k < len(ops):
                dur[k].append((e - s) / 1e3)
                names[k] = short[:60]
                k += 1
            else:
                other[short[:60]].append((e - s) / 1e3)
    pmc = defaultdict(lambda: defaultdict(list))   # op index -> counter -> values
    for db in a.pmc:
        cur = sqlite3.connect(db).cursor()
        rows = list(cur.execute("select dispatch_id, kernel_name, counter_name, value, start from counters_collection order by start, dispatch_id"))
        seq, seen = [], {}
        for did, kn, cn, v, s in rows:
            if did not in seen:
                seen[did] = len(seq)
                seq.append((kn, s, s + 1, {}))
            seq[seen[did]][3][cn] = v
        # same step splitting as for the trace, keeping the counter dicts
        starts = [i for i, d in enumerate(seq) if "conv_stem" in d[0] or "letterbox" in d[0] or "stem_body1" in d[0]]
        clean = []
        for i in starts:
            if clean and i - clean[-1] <= 2 and "letterbox" in seq[clean[-1]][0]:
                continue
            clean.append(i)
        clean = clean[-meta["steps"]:]
        for kk, i in enumerate(clean):
            j = clean[kk + 1] if kk + 1 < len(clean) else len(seq)
            k = 0
            for kn, _, _, cd in seq[i:j]:
                if any(t in kn for t in CONV_LIKE) and k < len(ops):
                    for cn, v in cd.items():
                        pmc[k][cn].append(v)
                    k += 1
    counters = sorted({c for d in pmc.values() for c in d})
    print("# per-launch table, " + meta["config"] + f", batch {meta['batch']}, median over {len(steps)} serial steps (one batch in flight); durations from rocprofv3 --kernel-trace; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (median duration x 2.4 GHz x 1024 SIMDs)")
    import csv
    wr = csv.writer(sys.stdout)
    wr.writerow(["op", "name", "shape", "tile", "kernel", "avg_us", "min_us", "tflops", "gbps", "bound_us", "frac_of_own_bound"] + counters + (["mfma_busy_frac"] if counters else []))
    tot_t = tot_b = 0.0
    for k, o in enumerate(ops):
        if not dur[k]:
            continue
        t = sorted(dur[k])[len(dur[k]) // 2]   # median: a shared box shows 2x outliers on single launches
        bound = sum(max(f / MFMA_PEAK, b / HBM_PEAK) for f, b in (o.get("layers") or [(o["flops"], o["bytes"])])) * 1e6   # one bound per reference conv
        tot_t += t
        tot_b += bound
        cvals = []
        for c in counters:
            v = pmc[k].get(c, [])
            cvals.append(f"{sum(v) / len(v):.0f}" if v else "")
        extra = []
        if counters:
            busy = pmc[k].get("SQ_VALU_MFMA_BUSY_CYCLES", [])
            gui = pmc[k].get("GRBM_GUI_ACTIVE", [])
            if busy:
                # SQ_VALU_MFMA_BUSY_CYCLES is summed over all 1024 SIMDs (= 32 x SQ_INSTS_MFMA for 32x32x16 f16).  GRBM_GUI_ACTIVE
                # (summed over the 8 XCDs) also counts ~15 us of dispatch bracket in counter mode, so it over-states short
                # kernels: the denominator is the kernel-trace duration at the 2.4 GHz peak clock
                denom = t * 1e-6 * 2.4e9 * 1024
                extra = [f"{(sum(busy) / len(busy)) / denom:.3f}"]
            else:
                extra = [""]
        wr.writerow([k, o["name"], o.get("shape", ""), o.get("tile", ""), names.get(k, ""), f"{t:.2f}", f"{min(dur[k]):.2f}", f"{o['flops'] / t / 1e6:.1f}", f"{o['bytes'] / t / 1e3:.1f}",
                     f"{bound:.2f}", f"{bound / t:.3f}"] + cvals + extra)
    print(f"# conv stack: sum of kernel durations {tot_t:.1f} us per step, sum of per-layer bounds {tot_b:.1f} us -> frac {tot_b / max(tot_t, 1e-9):.3f}")
    print("# other kernels of a step (avg us x launches per step):")
    for n, v in sorted(other.items(), key=lambda kv: -sum(kv[1])):
        print(f"#   {n}: {sorted(v)[len(v) // 2]:.2f} us x {len(v) / max(len(steps), 1):.1f}")


if __name__ == "__main__":
    main()
