"""dev: print kernel name / grid / duration (us) of every dispatch in a rocprofv3 kernel-trace directory matching a substring"""
import csv, glob, sys
d, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            print("%-60s grid %9s  %9.1f us" % (r["Kernel_Name"][:60], "%sx%s" % (r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", "")), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
