"""gpurun_out/calib/{true_bytes.txt, pmc_FETCH_SIZE, pmc_WRITE_SIZE} -> calibration.json: per access pattern the factor
(true bytes) / (counter bytes) of rocprofv3's FETCH_SIZE / WRITE_SIZE (KiB units) on gfx950."""
import collections, csv, glob, json, sys
from pathlib import Path
O = Path(sys.argv[1])
true = {}
for line in (O / "true_bytes.txt").read_text().splitlines():
    k, r, w = line.split()
    true[k] = (int(r), int(w))
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(str(O / "pmc_*" / "**" / "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        cnt[k][row["Counter_Name"]].append(float(row["Counter_Value"]) * 1024.0)
out = {}
for k, (r, w) in true.items():
    fs = cnt[k].get("FETCH_SIZE", [])
    ws = cnt[k].get("WRITE_SIZE", [])
    fetch = sum(fs) / len(fs) if fs else None
    write = sum(ws) / len(ws) if ws else None
    out[k] = dict(true_read_bytes=r, true_write_bytes=w, fetch_counter_bytes=fetch, write_counter_bytes=write,
                  fetch_factor=(r / fetch if (fetch and r) else None), write_factor=(w / write if (write and w) else None))
json.dump(out, open(O / "calibration.json", "w"), indent=1, sort_keys=True)
for k, v in out.items():
    print(f"{k:22s} read {v['true_read_bytes']/1e6:9.1f} MB counter {0 if v['fetch_counter_bytes'] is None else v['fetch_counter_bytes']/1e6:9.1f}  factor {v['fetch_factor']}"
          f" | write {v['true_write_bytes']/1e6:9.1f} MB counter {0 if v['write_counter_bytes'] is None else v['write_counter_bytes']/1e6:9.1f}  factor {v['write_factor']}")
