"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs (separate passes, csv) into profiles/pmc_latest.json.
Units: FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE
counts 64 B per 128-B request for wide (16 B/lane) coalesced streaming reads -> doubled for the kernels
whose reads are dominated by such streams; other kernels are reported raw (uncalibrated)."""
import collections, csv, json, sys
STREAMING = ("k_preprocess<", "k_preprocess_bwd<", "k_sh_eval<")   # wide (16 B/lane) streaming reads dominate
SLOT = {"k_preprocess<": "preprocess", "k_render<": "render", "k_render_bwd": "render_bwd", "k_preprocess_bwd<": "geometry_bwd",
        "k_sh_bwd<": "sh_bwd", "k_sh_eval<": "sh_eval", "k_render_bwd_em<": "render_bwd", "k_emit<": "emit", "k_gather_pairs": "gather_pairs", "k_cube2erp_fwd": "cube2erp"}


def slot_of(k):
    for pre, name in SLOT.items():
        if k.startswith(pre):
            return name
    return k
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[2:]:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("s360::", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, c in acc.items():
    if not k.startswith("k_"):
        continue
    fetch = sum(c.get("FETCH_SIZE", [0])) / max(len(c.get("FETCH_SIZE", [1])), 1) * 1024
    write = sum(c.get("WRITE_SIZE", [0])) / max(len(c.get("WRITE_SIZE", [1])), 1) * 1024
    corr = 2.0 if k.startswith(STREAMING) else 1.0
    out[slot_of(k)] = dict(kernel=k, fetch_bytes_raw=fetch, write_bytes_raw=write, fetch_correction=corr,
                               hbm_bytes_per_launch=fetch * corr + write)
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in out.items()}, indent=0))
