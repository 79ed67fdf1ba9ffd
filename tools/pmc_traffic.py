#!/usr/bin/env python3
"""Per-launch HBM traffic of the sparse-conv kernels from rocprofv3 PMC passes (FETCH_SIZE and
WRITE_SIZE collected in SEPARATE runs, as MI355X_MICROARCH.md prescribes: TCC has 4 counter slots,
FETCH_SIZE costs 3, WRITE_SIZE 2).  Units: both counters are reported in KiB by rocprofv3.
gfx950 correction (same guide, §HBM): FETCH_SIZE tallies 128-B fabric requests at 64 B, i.e. it reads
exactly 1/2 of the bytes of a wide coalesced stream -- so the read side is reported both raw and x2
(upper bound; the gather reads here are 64-B row segments, for which the factor is uncalibrated).
usage: pmc_traffic.py <fetch.db> <write.db> <out.json> [l2.db]"""
import collections
import json
import sqlite3
import sys


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, dispatch_id, sum(counter_value) from pmc_events where counter_name=? "
                       "group by name, dispatch_id", (counter,)).fetchall()
    agg = collections.defaultdict(list)
    for name, _, v in rows:
        agg[name.split("(")[0].replace("void ", "")].append(v)
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"units": "bytes per launch (average over all launches of the kernel symbol in `python bench.py "
                    "--steps 10 --warmup 3`)", "kernels": {}}
    l2 = {}
    if len(sys.argv) > 4:
        hit, miss = per_kernel(sys.argv[4], "TCC_HIT_sum"), per_kernel(sys.argv[4], "TCC_MISS_sum")
        l2 = {k: hit[k][0] / max(1.0, hit[k][0] + miss.get(k, (0, 0))[0]) for k in hit}
    for k in sorted(fetch):
        if "imf::" not in k:
            continue
        f_raw = fetch[k][0] * 1024.0
        w = write.get(k, (0.0, 0))[0] * 1024.0
        out["kernels"][k] = {"launches": fetch[k][1], "fetch_bytes_raw": round(f_raw), "write_bytes": round(w),
                             "hbm_bytes_raw": round(f_raw + w), "hbm_bytes_fetch_x2": round(2 * f_raw + w),
                             "l2_hit_rate": round(l2.get(k, float("nan")), 4)}
    with open(sys.argv[3], "w") as f:
        json.dump(out, f, indent=1)
    for k, v in out["kernels"].items():
        print(f"{k:44s} n={v['launches']:4d} fetch={v['fetch_bytes_raw'] / 1e6:9.2f} MB write={v['write_bytes'] / 1e6:8.2f} MB "
              f"L2hit={v['l2_hit_rate']}")


if __name__ == "__main__":
    main()
