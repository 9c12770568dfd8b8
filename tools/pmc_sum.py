"""Sum rocprofv3 counter_collection csvs per kernel: python tools/pmc_sum.py DIR [kernel-substring]."""
import csv, glob, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if sub in k:
            acc[k[:60]][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k[:60]][r["Counter_Name"]] += 1
for k in acc:
    print(k)
    for c in sorted(acc[k]):
        print("   %-32s %16.0f  per launch (%d)" % (c, acc[k][c] / cnt[k][c], cnt[k][c]))
