"""Where the path kernels' register spills sit (profiling helper, not a test; needs hipcc, no GPU).

    python3 tests/spill_report.py > profiles/r3_path_kernel_resources.txt

Compiles csrc/sppm_render.hip for gfx950 with the library's flags, reads the kernel resource remarks and the
device assembly, and prints per path kernel: registers, spills, scratch bytes per lane, LDS per block, and the
spill instructions (scratch_load / scratch_store) by the loop depth of the block they are in.  Depth 1 is the
kernel's turn loop (fetch work, walk, shade), depth 2 and deeper are the loops of the BVH walk."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "pathtracer_amd", "csrc", "sppm_render.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "pathtracer_amd", "csrc"), "-Wno-unused-function"]
WANT = ("k_camera_paths", "k_photon_paths", "k_camera_first", "k_photon_first")


def main():
    with tempfile.TemporaryDirectory() as tmp:
        # two compilations: the resource remarks are not printed when the temporaries are kept
        res = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-c",
                              "-o", os.path.join(tmp, "r.o"), SRC], capture_output=True, text=True, cwd=tmp)
        if res.returncode != 0:
            sys.exit(res.stderr[-2000:])
        res2 = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["--save-temps=obj", "-c", "-o", os.path.join(tmp, "o.o"), SRC],
                              capture_output=True, text=True, cwd=tmp)
        if res2.returncode != 0:
            sys.exit(res2.stderr[-2000:])
        asm = open([os.path.join(tmp, f) for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]).read()
    print("hipcc " + " ".join(f for f in FLAGS if not f.startswith("-I")) + "  (pathtracer_amd/csrc/sppm_render.hip)\n")
    print("%-46s %5s %6s %6s %8s %6s %5s" % ("kernel", "VGPR", "vspill", "sspill", "scratch", "LDS", "occ"))
    cur = {}
    for line in res.stderr.splitlines():
        m = re.search(r"remark: (.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
        elif ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
            if k.strip().startswith("LDS Size") and any(w in cur["name"] for w in WANT):
                name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip().split("(")[0]
                print("%-46s %5s %6s %6s %8s %6s %5s" % (name[-46:], cur.get("VGPRs"), cur.get("VGPRs Spill"), cur.get("SGPRs Spill"),
                                                       cur.get("ScratchSize [bytes/lane]"), cur.get("LDS Size [bytes/block]"),
                                                       cur.get("Occupancy [waves/SIMD]")))
    print("\nspill instructions by loop depth (static counts in the device assembly)")
    print("%-46s %5s %7s %8s %9s" % ("kernel", "depth", "instr", "sc_load", "sc_store"))
    for f in re.split(r"\n\t\.globl\t", asm):
        name = f.split("\n", 1)[0].strip()
        if not any(w in name for w in ("k_camera_paths", "k_photon_paths")):
            continue
        short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
        lines = f.split("\n")
        by = {}
        depth = 0
        for i, l in enumerate(lines):
            if re.match(r"^\.LBB\d+_\d+:", l):
                j, info = i + 1, l
                while j < len(lines) and lines[j].strip().startswith(";"):
                    info += lines[j]
                    j += 1
                d = re.findall(r"Depth=(\d+)", info)
                depth = max(map(int, d)) if d else 0
                continue
            if l.startswith("\t") and not l.startswith("\t.") and not l.strip().startswith(";"):
                b = by.setdefault(depth, [0, 0, 0])
                b[0] += 1
                b[1] += "scratch_load" in l
                b[2] += "scratch_store" in l
        for d in sorted(by):
            print("%-46s %5d %7d %8d %9d" % (short[-46:], d, *by[d]))


if __name__ == "__main__":
    main()
