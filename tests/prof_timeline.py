"""Prints the kernel timeline of one iteration from a rocprofv3 kernel trace csv (profiling helper, not a test)."""
import csv
import sys


def main(path, which=10, min_us=60.0):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    cams = [r for r in rows if "k_camera_paths" in r["Kernel_Name"]]
    t0 = int(cams[which]["Start_Timestamp"]) - 300000
    t1 = int(cams[which + 1]["Start_Timestamp"]) - 300000
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if t0 <= s < t1 and (e - s) > min_us * 1e3:
            print("%7.3f -> %7.3f  (%6.3f) q%-2s grid %-8s vgpr %-4s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6,
                  r["Queue_Id"], r["Grid_Size_X"], r["VGPR_Count"], r["Kernel_Name"][:64]))
    print("iteration span ms", (t1 - t0) / 1e6)


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10)
