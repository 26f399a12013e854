"""Per-unit timing of the backward composite (build with S360_HIPCC_EXTRA=-DS360_DBG_TIMING): is the kernel bound by its
longest (tile, quadrant) chain, by the tail of the schedule, or by throughput?"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from splatter360_amd import decoder, rasterizer, synthetic
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "encoder_like"
cloud = {"encoder_like": lambda: synthetic.encoder_like_cloud(512, 1024), "surface_like": lambda: synthetic.surface_like_cloud(512, 1024)}[name]()
rasterizer.SPLIT_LONG_LISTS = True if name == "surface_like" else "auto"
g = [torch.tensor(cloud[k], device=dev).requires_grad_(True) for k in ("means", "covariances", "harmonics", "opacities")]
ext, K, near, far = decoder.cube_cameras(torch.eye(4, device=dev), 0.1, 10.0)
for _ in range(3):
    faces = decoder.render_views_fused(ext, K, near, far, (256, 256), torch.zeros(3, device=dev), *g, check="lazy", shared_campos=True)
    st = rasterizer.last_state()
    st._arr(st.layout.keys_alt, 4 * (1536 * 4 + 65536), torch.int32).zero_()     # (the forward's own timing records live there)
    ((faces - 0.5) ** 2).mean().backward()
torch.cuda.synchronize()
nwork = int(st.header()[6].item()) if (st.prm.flags & 512) else 0      # segment units of split quadrants follow the (tile, quadrant) units
nu = 1536 * 4 + nwork
d = st._arr(st.layout.keys_alt, 4 * nu, torch.int32).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
print(name, "units:", 1536 * 4, "(tile, quadrant) +", nwork, "segment units of", int(st.header()[5].item()) if nwork else 0, "split quadrants")
t0, dur, wl, blk = d[0::4], d[1::4], d[2::4], d[3::4]
ok = dur > 0
tick = 100.0
start = ((t0 - t0[ok].min()) & 0xFFFFFFFF) / tick
dur = dur / tick
end = start + dur
print("units timed", ok.sum(), "of", nu, " kernel span us", end[ok].max())
print("unit duration mean/median/p90/p99/max", dur[ok].mean(), np.median(dur[ok]), np.percentile(dur[ok], 90), np.percentile(dur[ok], 99), dur[ok].max())
print("sum of durations / 3072 wave slots us:", dur[ok].sum() / 3072, " longest unit / mean unit:", dur[ok].max() / dur[ok].mean())
print("start pct [50,90,99,max]", np.percentile(start[ok], [50, 90, 99, 100]))
halves, surv = blk >> 16, blk & 0xFFFF
print("survivors per unit mean", surv[ok].mean(), "survivor fraction", surv[ok].sum() / wl[ok].sum(), "four-pixel runs executed per unit mean", halves[ok].mean(), "per group", halves[ok].sum() / np.maximum(np.ceil(surv[ok] / 64), 1).sum())
print("walk length mean/max", wl[ok].mean(), wl[ok].max(), " us per 1000 walked entries (median)", np.median(dur[ok] / np.maximum(wl[ok], 1) * 1000))
o = np.argsort(-end)[:8]
for i in o:
    print("unit", i, "blk", blk[i], "start", round(start[i], 1), "dur", round(dur[i], 1), "walk", wl[i])
o = np.argsort(-dur)[:8]
for i in o:
    print("longest: unit", i, "blk", blk[i], "start", round(start[i], 1), "dur", round(dur[i], 1), "walk", wl[i])
# concurrency over time
ts = np.linspace(0, end[ok].max(), 25)
print("running units at t:", [(round(t), int(((start <= t) & (end > t) & ok).sum())) for t in ts])
