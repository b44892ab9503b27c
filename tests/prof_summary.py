"""Prints per-kernel summaries of rocprofv3 csv output (kernel stats, or a --pmc counter collection).  A helper for
profiling sessions, not a test."""
import collections
import csv
import sys


def stats(path, top=24):
    rows = list(csv.DictReader(open(path)))
    for r in rows[:top]:
        print("%-80s %6s %10.1f %9.1f %5.1f" % (r["Name"][:80], r["Calls"], float(r["TotalDurationNs"]) / 1e3,
                                                  float(r["AverageNs"]) / 1e3, float(r["Percentage"])))


def pmc(path, top=16):
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    seen = set()
    for r in rows:
        k = r["Kernel_Name"][:56]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            cnt[k] += 1
    names = sorted({r["Counter_Name"] for r in rows})
    print("%-56s %5s " % ("kernel", "disp") + " ".join("%14s" % n[-14:] for n in names))
    for k, c in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:top]:
        print("%-56s %5d " % (k, cnt[k]) + " ".join("%14.4g" % (c[n] / cnt[k]) for n in names))


if __name__ == "__main__":
    (pmc if sys.argv[1] == "pmc" else stats)(sys.argv[2])
