"""Developer helper: mean of each PMC counter per kernel from a rocprofv3 --pmc output dir."""
import csv, glob, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        m = re.search(r"(\w+_(image|plane)\w*_kernel)", k) or (len(sys.argv) > 2 and re.search(r"((rle|rld|sym|hd|scan)\w*_kernel)", k))
        if not m:
            continue
        acc[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
