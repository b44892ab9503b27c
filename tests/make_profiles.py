"""Turns the raw output of a profiling session (bench lines, rocprofv3 kernel stats and --pmc passes, see the
command list printed with --help) into the summaries kept under profiles/.  A helper, not a test.

    python3 tests/make_profiles.py <session dir> <round tag, e.g. r1>

The session directory holds: line_bench.json, line_bench_steps20_warmup5.json, line_bench_under_rocprof.json (bench lines),
kt/ (rocprofv3 --kernel-trace --stats), sq/ (--pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU), fs/ (--pmc FETCH_SIZE), ws/ (--pmc WRITE_SIZE), every pass
a run of `python3 bench.py --no-cpu-baseline` of its own."""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def load_pmc(d):
    path = glob.glob(os.path.join(d, "*counter_collection.csv"))[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        seen[k].add(r["Dispatch_Id"])
    return agg, {k: len(v) for k, v in seen.items()}


def main(src, tag):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "profiles")
    for name in ("bench.json", "bench_steps20_warmup5.json", "bench_under_rocprof.json"):
        line = [l for l in open(os.path.join(src, "line_" + name)) if l.startswith("{")][-1]
        with open(os.path.join(out, "%s_%s" % (tag, name)), "w") as f:
            f.write(json.dumps(json.loads(line), indent=1) + "\n")
    shutil.copy(glob.glob(os.path.join(src, "kt", "*kernel_stats.csv"))[0], os.path.join(out, tag + "_bench_kernel_stats.csv"))

    agg, disp = load_pmc(os.path.join(src, "sq"))
    with open(os.path.join(out, tag + "_pmc_sq_summary.txt"), "w") as f:
        f.write("SQ counters per kernel (rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD "
                "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU; `python3 bench.py --no-cpu-baseline`)\n")
        f.write("%-56s %5s %9s %11s %9s %8s %7s %8s %6s\n" % ("kernel", "disp", "waves/d", "VALU/d", "VALU/w", "SALU/w", "LDS/w", "VMEM/w", "lane%"))
        for k, c in sorted(agg.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU"])[:16]:
            w = max(c["SQ_WAVES"], 1.0)
            lane = 100.0 * c["SQ_THREAD_CYCLES_VALU"] / max(64.0 * c["SQ_ACTIVE_INST_VALU"], 1.0)
            f.write("%-56s %5d %9.0f %11.3e %9.0f %8.0f %7.0f %8.0f %6.1f\n" % (
                k[:56], disp[k], c["SQ_WAVES"] / disp[k], c["SQ_INSTS_VALU"] / disp[k], c["SQ_INSTS_VALU"] / w,
                c["SQ_INSTS_SALU"] / w, c["SQ_INSTS_LDS"] / w, c["SQ_INSTS_VMEM_RD"] / w, lane))

    per = {}
    with open(os.path.join(out, tag + "_pmc_hbm_traffic.txt"), "w") as f:
        for d, name in (("fs", "FETCH_SIZE"), ("ws", "WRITE_SIZE")):
            agg, disp = load_pmc(os.path.join(src, d))
            f.write("%s (rocprofv3 --pmc %s, a pass of its own; KB; `python3 bench.py --no-cpu-baseline`)\n" % (name, name))
            for k, c in sorted(agg.items(), key=lambda kv: -kv[1][name])[:12]:
                f.write("   %-60s dispatches %4d  total %.4g  per dispatch %.4g\n" % (k[:60], disp[k], c[name], c[name] / disp[k]))
                if "k_gather_group" in k:
                    per[name] = c[name] / disp[k]
                if "k_camera_paths<false>" in k:
                    per["camera_" + name] = c[name] / disp[k]
    fetch, write = per["FETCH_SIZE"], per["WRITE_SIZE"]
    cam_fetch, cam_write = per.get("camera_FETCH_SIZE", 0.0), per.get("camera_WRITE_SIZE", 0.0)
    with open(os.path.join(out, "hbm_traffic.json"), "w") as f:
        json.dump({"kernel": "k_gather_group", "fetch_size_kb_per_launch": fetch, "write_size_kb_per_launch": write,
                   "bytes_per_launch": (2.0 * fetch + write) * 1024.0,
                   "k_camera_paths_fetch_size_kb_per_launch": cam_fetch, "k_camera_paths_write_size_kb_per_launch": cam_write,
                   "k_camera_paths_bytes_per_launch": (2.0 * cam_fetch + cam_write) * 1024.0,
                   "collected": tag,
                   "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, averaged over the launches "
                           "(one launch per iteration takes the groups of both maps); FETCH_SIZE doubled (gfx950 tallies the "
                           "128-byte requests of wide coalesced reads at 64 bytes); uncorrected: %d bytes"
                           % int((fetch + write) * 1024.0)}, f, indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
