"""Sum rocprofv3 SQ counters per kernel (encode passes A / B told apart by workgroup size).  usage: sq_counters.py <counter_collection.csv> ..."""
import collections, csv, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        full = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hobbit::", "")
        name = full.split("<")[0]
        if name == "k_enc_fat":
            name = "k_enc_fat_" + {"2": "A", "1": "C1", "3": "D"}.get(full.split("<")[1][:1], "?")
        elif name.startswith("k_encode"):
            name = "k_encode_A" if int(r["Workgroup_Size"]) > 512 else "k_encode_M2" if int(r["Workgroup_Size"]) <= 128 else "k_encode_B"
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(agg.items()):
    if any(x in k for x in ("encode", "enc_fat", "fft4096", "leaf_chain", "transpose")) and "tw" not in k:
        print(k, {a: "%.3g" % b for a, b in sorted(v.items())})
