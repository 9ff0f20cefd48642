import csv, glob, sys, collections
for d in sys.argv[1:]:
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_f64_kernel" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    print(d, {k: (v[0], "%.3f GB" % (v[1] * 1024 * (2 if k == "FETCH_SIZE" else 1) / 1e9)) for k, v in acc.items()})
