#!/usr/bin/env python3
"""Where the backward composite's evaluated slots go, per workload at 1 M Gaussians, six 256^2 faces:
profiles/r06_bwd_slot_breakdown.txt.  Usage: bwd_slots.py [encoder_like|surface_like|uniform ...]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from splatter360_amd import decoder, rasterizer, synthetic

dev = torch.device("cuda:0")
rasterizer.SPLIT_LONG_LISTS = False
ext, K, near, far = decoder.cube_cameras(torch.eye(4, device=dev), 0.1, 10.0)
bg = torch.zeros(3, device=dev)
for name in (sys.argv[1:] or ["encoder_like", "surface_like", "uniform"]):
    cloud = {"encoder_like": lambda: synthetic.encoder_like_cloud(512, 1024, n_context=2, d_sh=25, seed=0),
             "surface_like": lambda: synthetic.surface_like_cloud(512, 1024, n_context=2, seed=0),
             "uniform": lambda: synthetic.uniform_cloud(1 << 20, seed=0, extent=5.0)}[name]()
    ps = [torch.tensor(cloud[k], device=dev, requires_grad=True) for k in ("means", "covariances", "harmonics", "opacities")]
    views = decoder.pack_camera_views(ext, K, near, far, bg)
    faces = decoder.render_views_fused(ext, K, near, far, (256, 256), bg, *ps, shared_campos=True, views=views)
    st = rasterizer.last_state()
    r = st.count_backward_slots()
    e = r["executed"]
    print(f"== {name}: {st.num_rendered()} instances, {r['units']} units, {r['groups']} groups of <= 64 records")
    print(f"   slots executed {e:,} (+ {r['skipped_runs']:,} in runs the reach test skipped = {r['skipped_runs'] / (e + r['skipped_runs']):.1%} of all)")
    for k in ("contributing", "miss", "stopped", "padding"):
        print(f"   {k:13s} {r[k]:>16,}  {r[k] / e:6.1%}")
    print("   executed slots by the unit's contributing fraction (deciles 0-10% ... 90-100%):", " ".join(f"{x / e:.1%}" for x in r["by_unit_hit_decile"]))
    tot_runs = sum(r["runs_by_active_records"].values())
    print("   executed four-pixel runs by records (of 64 lanes) contributing to the run:", {k: f"{v / tot_runs:.1%}" for k, v in r["runs_by_active_records"].items()})
    n = r["records"]
    print(f"   records {n:,}: reach the quadrant's upper 8x4 half {r['reach_upper_half'] / n:.1%}, lower {r['reach_lower_half'] / n:.1%}, both {r['reach_both_halves'] / n:.1%};"
          f" a half-wave composite (32 + 32 lanes, 8 runs per iteration) would execute {r['half_wave_iterations'] * 8:,} runs against {e // 256:,} today")
