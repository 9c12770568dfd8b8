"""Timeline of one replayed train_dnn step from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace -d /tmp/tl -o t --output-format csv -- python tools/time_train_dnn_graph.py 4200
    python tools/step_timeline.py /tmp/tl [n_steps_from_the_end]
Prints, for the last complete steps, every kernel's start (us after the step's first kernel), duration and queue."""
import csv, glob, sys
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("void ", "").replace("pinn::x6::", "").replace("pinn::", "").replace("(anonymous namespace)::", "")[:60]
# a step starts at the pack kernel
idx = [i for i, r in enumerate(rows) if "pack_x6_kernel" in r["Kernel_Name"]]
for a, b in list(zip(idx[:-1], idx[1:]))[-back:]:
    t0 = int(rows[a]["Start_Timestamp"])
    print("--- step: %.1f us to the next step's first kernel" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("  +%7.1f  %6.1f us  q%-3s %s  grid %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), short(r["Kernel_Name"]),
                                                     r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
