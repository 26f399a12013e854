"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs (separate passes, csv) into profiles/pmc_latest.json.
Units: FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction, CALIBRATED on this pool (scripts/calib/, result committed as
profiles/fetch_write_calibration.json): FETCH_SIZE reports exactly half the bytes the memory side delivers for EVERY read
pattern this library uses (16-byte and 4-byte coalesced streams, 48-byte record streams, non-temporal loads: factor 2.000; a
random 48-byte gather reports 1.6x its useful bytes = 0.5 x the 3.2x line over-fetch it really causes), so every kernel's
FETCH_SIZE is doubled; WRITE_SIZE is exact (streaming stores 1.000; scattered 48-byte records 1.35x their payload and scattered
single bytes 32 B each are real write amplification, not counter error)."""
import collections, csv, json, sys
FETCH_CORRECTION = 2.0
SLOT = {"k_preprocess<": "preprocess", "k_render<": "render", "k_preprocess_bwd<": "preprocess_bwd", "k_sh_bwd<": "sh_bwd",
        "k_sh_eval": "sh_eval", "k_render_bwd_em<": "render_bwd", "k_emit<": "emit", "k_gather_slots": "gather_slots",
        "k_cube2erp_fwd": "cube2erp", "k_sort_stage1": "sort_tiles"}


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
steps = max([len(c.get("FETCH_SIZE", [])) for k, c in acc.items() if k.startswith("k_render<")] + [1])   # one k_render per step
total = 0.0
for k, c in acc.items():
    if not k.startswith("k_"):
        continue
    fetch = sum(c.get("FETCH_SIZE", [0])) / max(len(c.get("FETCH_SIZE", [1])), 1) * 1024
    write = sum(c.get("WRITE_SIZE", [0])) / max(len(c.get("WRITE_SIZE", [1])), 1) * 1024
    launches = max(len(c.get("FETCH_SIZE", [])), len(c.get("WRITE_SIZE", [])))
    rec = dict(kernel=k, fetch_bytes_raw=fetch, write_bytes_raw=write, fetch_correction=FETCH_CORRECTION,
               hbm_bytes_per_launch=fetch * FETCH_CORRECTION + write, launches_per_step=launches / steps)
    total += rec["hbm_bytes_per_launch"] * rec["launches_per_step"]
    out[slot_of(k) if slot_of(k) not in out else k] = rec
out["_meta"] = dict(counter_bytes_per_step=total, steps_profiled=steps)
json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in out.items() if k != "_meta"}, indent=0), "per step MB:", round(total / 1e6, 1))
